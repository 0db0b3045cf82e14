// ubench_xstream.hip — cost of a cross-stream dependency inside a chain of short kernels (measurement tool only).
// Pattern per step (the overlapped run of amc_api.hip):  main: A -> [signal] -> B -> [wait side] -> C ;  side: [wait signal] -> P -> [signal back]
// Variants: 0 = everything on one stream (no dependency), 1 = hipEvent record / wait, 2 = hipStreamWriteValue32 / hipStreamWaitValue32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void k_work(double *x, int n, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    for (int k = 0; k < iters; k++) v = fma(v, 1.0000001, 1e-9);
    x[i] = v;
}

int main()
{
    const int n = 1 << 20;
    double *a, *b;
    hipMalloc(&a, n * sizeof(double));
    hipMalloc(&b, n * sizeof(double));
    hipMemset(a, 0, n * sizeof(double));
    hipMemset(b, 0, n * sizeof(double));
    unsigned int *flag;
    if (hipExtMallocWithFlags((void **)&flag, 256, hipMallocSignalMemory) != hipSuccess) { printf("signal memory: no\n"); hipMalloc(&flag, 256); }
    hipMemset(flag, 0, 64);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreateWithFlags(&e0, hipEventDisableTiming);
    hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    const int steps = 2000;
    for (int iters : {50, 2000}) {
        for (int variant = 0; variant < 3; variant++) {
            if (variant == 2 && !can) continue;
            unsigned int tick = 0;
            hipMemset(flag, 0, 64);
            hipDeviceSynchronize();
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int s = 0; s < steps; s++) {
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s0, a, n, iters);                 // A (detect)
                hipStream_t side = variant == 0 ? s0 : s1;
                tick++;
                if (variant == 1) { hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0); }
                if (variant == 2) { CK(hipStreamWriteValue32(s0, flag, tick, 0)); CK(hipStreamWaitValue32(s1, flag, tick, hipStreamWaitValueGte, 0xffffffffu)); }
                hipLaunchKernelGGL(k_work, dim3(64), dim3(64), 0, s0, a, 4096, iters * 8);                 // B (latency-bound resolve)
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, side, b, n, iters);              // P (the pass)
                if (variant == 1) { hipEventRecord(e1, s1); hipStreamWaitEvent(s0, e1, 0); }
                if (variant == 2) { CK(hipStreamWriteValue32(s1, flag + 16, tick, 0)); CK(hipStreamWaitValue32(s0, flag + 16, tick, hipStreamWaitValueGte, 0xffffffffu)); }
                hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s0, a, 16384, iters);                   // C (fix-up)
            }
            hipDeviceSynchronize();
            if (variant == 2) { unsigned int h[32]; hipMemcpy(h, flag, sizeof h, hipMemcpyDeviceToHost); printf("  flags after the run: %u %u (expected %u)\n", h[0], h[16], tick); }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / steps;
            printf("iters %4d variant %d (%s): %.2f us per step\n", iters, variant,
                   variant == 0 ? "one stream" : (variant == 1 ? "events" : "stream write/wait value"), us);
        }
    }
    return 0;
}
