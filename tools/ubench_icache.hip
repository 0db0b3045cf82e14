// ubench_icache.hip — cost of COLD instruction fetch for straight-line code executed once per launch by a lone wave (the
// situation of the resolve kernels), and whether a kernel's code survives in the instruction cache from one launch to the
// next.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int N>
__device__ __forceinline__ double body(double x, double y)
{
    // N independent-ish fp64 FMAs, fully unrolled: 8 bytes of code each
#pragma unroll
    for (int i = 0; i < N; i++) x = fma(x, y, (double)(i & 7) + 0.5);
    return x;
}

template <int N>
__global__ __launch_bounds__(64) void k_code(double *out, long long *t, int passes)
{
    double x = out[0], y = out[1];
    for (int p = 0; p < passes; p++) {
        const long long t0 = wall_clock64();
        x = body<N>(x, y);
        const long long t1 = wall_clock64();
        if (threadIdx.x == 0 && p < 4) atomicAdd((unsigned long long *)&t[p], (unsigned long long)(t1 - t0));
        y += 1e-9;
    }
    if (x == 1.2345) out[2] = x;
}

__global__ void k_other(double *out) { if (out[0] == 9.9) out[3] = 1; }

template <int N>
void run(const char *name, double *out, long long *t, int blocks)
{
    for (int variant = 0; variant < 3; variant++) {
        hipMemset(t, 0, 64);
        const int reps = 50;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL((k_code<N>), dim3(blocks), dim3(64), 0, 0, out, t, 2);
            if (variant == 1) hipLaunchKernelGGL(k_other, dim3(1024), dim3(256), 0, 0, out);
            if (variant == 2) hipDeviceSynchronize();
        }
        long long h[4];
        hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
        printf("%s (%d B of code), %3d blocks, %s: first pass %.2f us, second pass %.2f us\n", name, N * 8, blocks,
               variant == 0 ? "back to back      " : (variant == 1 ? "other kernel between" : "host sync between  "),
               h[0] / 100.0 / reps / blocks, h[1] / 100.0 / reps / blocks);
    }
}

int main()
{
    double *out;
    long long *t;
    hipMalloc(&out, 64);
    hipMalloc(&t, 64);
    hipMemset(out, 0, 64);
    for (int blocks : {1, 256, 512}) {
        run<512>("N=512 ", out, t, blocks);
        run<2048>("N=2048", out, t, blocks);
        run<8192>("N=8192", out, t, blocks);
    }
    return 0;
}
