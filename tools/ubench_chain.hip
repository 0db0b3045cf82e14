// ubench_chain.hip — what ONE dependent memory round trip costs a lone wave right after a kernel boundary, on data the
// previous kernel wrote from all CUs (the situation of every phase of the resolve kernels).  Measurement tool only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/ubench_chain && tools/ubench_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k_write(int *idx, const int *perm, int m, int use_atomic)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (use_atomic) atomicExch(&idx[i], perm[i]); else idx[i] = perm[i];
}

template <int HOPS>
__global__ __launch_bounds__(64) void k_chain(const int *idx, int m, int lanes, long long *out, int agent)
{
    const int lane = threadIdx.x;
    int p = (int)(((long long)blockIdx.x * 7919 + lane * 104729) % m);
    const long long t0 = wall_clock64();
    if (lane < lanes) {
#pragma unroll 1
        for (int h = 0; h < HOPS; h++)
            p = agent ? __hip_atomic_load(&idx[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : idx[p];
    }
    const long long t1 = wall_clock64();
    if (p == -12345) out[0] = 1;
    if (lane == 0) atomicAdd((unsigned long long *)&out[1], (unsigned long long)(t1 - t0));
}

int main()
{
    const int sizes[] = {1 << 15, 1 << 20, 1 << 24};   // ints: 128 KB, 4 MB, 64 MB
    long long *out;
    hipMalloc(&out, 64);
    for (int m : sizes) {
        std::vector<int> perm(m);
        for (int i = 0; i < m; i++) perm[i] = (int)(((long long)i * 1103515245LL + 12345) % m);
        int *d_idx, *d_perm;
        hipMalloc(&d_idx, sizeof(int) * m);
        hipMalloc(&d_perm, sizeof(int) * m);
        hipMemcpy(d_perm, perm.data(), sizeof(int) * m, hipMemcpyHostToDevice);
        for (int use_atomic = 0; use_atomic < 2; use_atomic++)
            for (int agent = 0; agent < 2; agent++)
                for (int blocks : {1, 64, 512})
                    for (int lanes : {1, 64}) {
                        double tot = 0;
                        const int reps = 20;
                        for (int r = 0; r < reps; r++) {
                            hipMemset(out, 0, 64);
                            hipLaunchKernelGGL(k_write, dim3((m + 255) / 256), dim3(256), 0, 0, d_idx, d_perm, m, use_atomic);
                            hipLaunchKernelGGL((k_chain<8>), dim3(blocks), dim3(64), 0, 0, d_idx, m, lanes, out, agent);
                            long long h[2];
                            hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
                            tot += (double)h[1] / blocks / 8.0 / 100.0;
                        }
                        printf("ints %9d written by %s, %s loads, %3d blocks x %2d lanes: %.3f us per hop\n", m,
                               use_atomic ? "atomics" : "stores ", agent ? "agent" : "plain", blocks, lanes, tot / reps);
                    }
        hipFree(d_idx);
        hipFree(d_perm);
    }
    return 0;
}
