// ubench_firsthops.hip — latency of the 1st, 2nd, 3rd ... dependent memory batch of a lone wave, measured from the wave's
// start, right after a kernel boundary.  Question: is there a window after kernel start in which requests stall
// (k_clusters_wide sees its SECOND batch take ~4 us whatever it reads, the first and third ~0.6 us)?
// Measurement tool only.   hipcc --offload-arch=gfx950 -O3 tools/ubench_firsthops.hip -o /tmp/ubench_firsthops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define HOPS 8
__global__ void k_write(int *idx, const int *perm, int m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) idx[i] = perm[i];
}

struct big_args { long long pad[100]; const int *idx; long long *out; int m, sleep_ticks, flat, scratch; };

// sleep_ticks: wait that long (100 MHz ticks) after the first hop; flat: the pointer takes a detour through LDS (generic
// address space: flat_load instead of global_load); scratch: the wave uses private memory (a dynamically indexed array)
__global__ __launch_bounds__(64) void k_hops(big_args A_in)
{
    __shared__ big_args s_args;
    {
        const int *src = (const int *)__builtin_amdgcn_kernarg_segment_ptr();
        int *dst = (int *)&s_args;
        for (int i = threadIdx.x; i < (int)(sizeof(big_args) / 4); i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const big_args &A = s_args;
    const int lane = threadIdx.x;
    const int *idx = A.flat ? A.idx : A_in.idx;
    const int m = A_in.m;
    int p = (int)(((long long)blockIdx.x * 7919 + lane * 104729) % m);
    long long t[HOPS + 1];
    int priv[16];
    if (A_in.scratch) for (int i = 0; i < 16; i++) priv[(i * 5 + lane) & 15] = i;
    t[0] = wall_clock64();
#pragma unroll
    for (int h = 0; h < HOPS; h++) {
        p = idx[p];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p == -12345) A_in.out[0] = 1;
        if (A_in.scratch) p = (p + (priv[(p + h) & 15] & 0)) ;
        t[h + 1] = wall_clock64();
    }
    if (lane == 0)
        for (int h = 0; h < HOPS; h++) atomicAdd((unsigned long long *)&A_in.out[8 + h + 16 * (blockIdx.x & 3)], (unsigned long long)(t[h + 1] - t[h]));
}

__global__ void k_atomics(unsigned long long *tab, int m, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicExch(&tab[(int)(((long long)i * 2654435761LL) % m)], (unsigned long long)i);
}

int main()
{
    const int m = 1 << 22;      // 16 MB of ints
    long long *out;
    hipMalloc(&out, 1024);
    std::vector<int> perm(m);
    for (int i = 0; i < m; i++) perm[i] = (int)(((long long)i * 1103515245LL + 12345) % m);
    int *d_idx, *d_perm;
    hipMalloc(&d_idx, sizeof(int) * m);
    hipMalloc(&d_perm, sizeof(int) * m);
    hipMemcpy(d_perm, perm.data(), sizeof(int) * m, hipMemcpyHostToDevice);
    unsigned long long *d_tab;
    hipMalloc(&d_tab, sizeof(unsigned long long) * (1 << 20));
    for (int pre : {1, 2})
        for (int blocks : {128, 512})
            for (int flat : {0, 1})
                for (int scratch : {0, 1}) {
                    double hop[HOPS] = {0};
                    const int reps = 20;
                    for (int r = 0; r < reps; r++) {
                        hipMemset(out, 0, 1024);
                        hipLaunchKernelGGL(k_write, dim3((m + 255) / 256), dim3(256), 0, 0, d_idx, d_perm, m);
                        if (pre == 2) hipLaunchKernelGGL(k_atomics, dim3(400), dim3(256), 0, 0, d_tab, 1 << 20, 100000);
                        big_args A;
                        A.idx = d_idx; A.out = out; A.m = m; A.sleep_ticks = 0; A.flat = flat; A.scratch = scratch;
                        hipLaunchKernelGGL(k_hops, dim3(blocks), dim3(64), 0, 0, A);
                        long long h[128];
                        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
                        for (int q = 0; q < HOPS; q++) hop[q] += (double)(h[8 + q] + h[24 + q] + h[40 + q] + h[56 + q]) / blocks / 100.0;
                    }
                    printf("after %s, %4d blocks, %s loads, %s: hops", pre == 2 ? "writer + atomics kernels" : "a writer kernel         ", blocks, flat ? "flat  " : "global",
                           scratch ? "private memory" : "no private mem");
                    for (int q = 0; q < HOPS; q++) printf(" %.2f", hop[q] / reps);
                    printf(" us\n");
                }
    return 0;
}
