// ubench_kernarg.hip — what it costs a lone wave to read a LARGE by-value kernel argument (kernarg segment) field by
// field, against the same structure read through a pointer to device memory.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>

struct big_args { int *p[160]; };      // 1280 bytes = 20 cache lines of kernarg

__global__ __launch_bounds__(64) void k_byval(big_args A, long long *t, int nfields)
{
    const long long t0 = wall_clock64();
    int acc = 0;
    // dependent chain: the value read through one pointer selects nothing, but the loads are ordered by a data dependence
    for (int f = 0; f < nfields; f++) {
        const int *q = A.p[(f * 8 + (acc & 0)) % 160];      // a new 64-byte line of the argument block every step
        acc += q[threadIdx.x & 1];
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { atomicAdd((unsigned long long *)&t[0], (unsigned long long)(t1 - t0)); if (acc == -7) t[3] = 1; }
}

__global__ __launch_bounds__(64) void k_byptr(const big_args *Ap, long long *t, int nfields)
{
    const long long t0 = wall_clock64();
    int acc = 0;
    for (int f = 0; f < nfields; f++) {
        const int *q = Ap->p[(f * 8 + (acc & 0)) % 160];
        acc += q[threadIdx.x & 1];
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { atomicAdd((unsigned long long *)&t[0], (unsigned long long)(t1 - t0)); if (acc == -7) t[3] = 1; }
}

int main()
{
    long long *t;
    int *data;
    big_args h, *d;
    hipMalloc(&t, 64);
    hipMalloc(&data, 4096);
    hipMemset(data, 0, 4096);
    for (int i = 0; i < 160; i++) h.p[i] = data + (i % 64) * 16;
    hipMalloc(&d, sizeof h);
    hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
    for (int blocks : {1, 256}) {
        for (int nf : {1, 4, 16}) {
            const int reps = 50;
            hipMemset(t, 0, 64);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_byval, dim3(blocks), dim3(64), 0, 0, h, t, nf);
            long long v[4];
            hipMemcpy(v, t, sizeof v, hipMemcpyDeviceToHost);
            printf("by value  : %3d blocks, %2d fields (one per 64-B line): %.2f us per wave\n", blocks, nf, v[0] / 100.0 / reps / blocks);
            hipMemset(t, 0, 64);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_byptr, dim3(blocks), dim3(64), 0, 0, d, t, nf);
            hipMemcpy(v, t, sizeof v, hipMemcpyDeviceToHost);
            printf("by pointer: %3d blocks, %2d fields (one per 64-B line): %.2f us per wave\n", blocks, nf, v[0] / 100.0 / reps / blocks);
        }
    }
    return 0;
}
