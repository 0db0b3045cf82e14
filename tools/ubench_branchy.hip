// ubench_branchy.hip — what does a lone wave pay for a TAKEN branch to code it has not fetched yet?  (The resolve kernels are
// 100+ KB of code each, executed once per launch by one wave per CU along a path full of far branches.)  ISLANDS pieces of 8
// chained fp64 FMAs, each followed by a branch over PAD bytes of never-executed code to the next piece; first pass (cold)
// against second pass (the path is in the instruction cache if it fits).  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define STR2(x) #x
#define STR(x) STR2(x)
#define PIECE(PADW)                                                                                   \
    "v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n" \
    "v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %0, %0, %1, %0\n" \
    "s_branch 1f\n .fill " STR(PADW) ", 4, 0xbf800000\n 1:\n"
#define P4(W) PIECE(W) PIECE(W) PIECE(W) PIECE(W)
#define P16(W) P4(W) P4(W) P4(W) P4(W)

template <int PADW>
__global__ __launch_bounds__(64) void k_br(double *out, long long *t, int passes)
{
    double x = out[0], y = out[1];
    for (int p = 0; p < passes; p++) {
        const long long t0 = wall_clock64();
        if (PADW == 16) asm volatile(P16(16) : "+v"(x) : "v"(y));
        else if (PADW == 256) asm volatile(P16(256) : "+v"(x) : "v"(y));
        else asm volatile(P16(1024) : "+v"(x) : "v"(y));
        const long long t1 = wall_clock64();
        if (threadIdx.x == 0 && p < 4) atomicAdd((unsigned long long *)&t[p], (unsigned long long)(t1 - t0));
        y += 1e-9;
    }
    if (x == 1.2345) out[2] = x;
}
__global__ void k_other(double *out) { if (out[0] == 9.9) out[3] = 1; }

template <int PADW>
void run(double *out, long long *t, int blocks)
{
    for (int variant = 0; variant < 2; variant++) {
        hipMemset(t, 0, 64);
        const int reps = 50;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL((k_br<PADW>), dim3(blocks), dim3(64), 0, 0, out, t, 3);
            if (variant == 1) hipLaunchKernelGGL(k_other, dim3(1024), dim3(256), 0, 0, out);
        }
        long long h[4];
        hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
        printf("16 pieces, %5d B between them (%3d KB of code), %3d blocks, %s: pass 1 %.2f us, pass 2 %.2f, pass 3 %.2f  => per cold branch %.0f ns, warm %.0f ns\n",
               PADW * 4, 16 * (PADW * 4 + 68) / 1024, blocks, variant == 0 ? "back to back        " : "other kernel between",
               h[0] / 100.0 / reps / blocks, h[1] / 100.0 / reps / blocks, h[2] / 100.0 / reps / blocks,
               (h[0] - h[2]) * 10.0 / reps / blocks / 16, h[2] * 10.0 / reps / blocks / 16);
    }
}

int main()
{
    double *out; long long *t;
    hipMalloc(&out, 64); hipMalloc(&t, 64); hipMemset(out, 0, 64);
    for (int blocks : {1, 512}) {
        run<16>(out, t, blocks);
        run<256>(out, t, blocks);
        run<1024>(out, t, blocks);
    }
    return 0;
}
