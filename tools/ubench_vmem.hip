// Micro-benchmark: what does a vector-memory load instruction cost when only a few of its 64 lanes are active?
// Each wave issues ITER loads of 8 or 32 bytes from pseudo-random addresses of a large table; `active` lanes take part.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES>
__global__ __launch_bounds__(256) void k(const double4 *tab, size_t mask, int iters, int active, int sorted, double *out)
{
    const int lane = threadIdx.x & 63;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t a = gid * 0x9E3779B97F4A7C15ULL;
    double acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            a = a * 6364136223846793005ULL + 1442695040888963407ULL;
            size_t idx = sorted ? ((gid * 4 + (size_t)i * 1048576 * 4) & mask) : ((a >> 20) & mask);
            if (BYTES == 32) { const double4 v = tab[idx]; acc += v.x + v.w; }
            else { acc += ((const double *)tab)[idx * 4]; }
        }
    }
    if (acc == 1234.5) out[0] = acc;
}

int main()
{
    const size_t n = (size_t)1 << 22;            // 4M records x 32 B = 128 MB
    double4 *tab; double *out;
    hipMalloc(&tab, n * sizeof(double4)); hipMemset(tab, 0, n * sizeof(double4)); hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 3907, iters = 16;         // 1e6 lanes
    for (int bytes : {8, 32})
        for (int sorted : {0, 1})
            for (int active : {64, 32, 16, 4, 1}) {
                float best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    if (bytes == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, iters, active, sorted, out);
                    else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, iters, active, sorted, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                const double winstr = (double)blocks * 4 * iters;
                printf("bytes=%2d sorted=%d active=%2d : %8.1f us  %6.1f ns/wave-instr  lane-req/s=%.3g\n", bytes, sorted, active,
                       best * 1e3, best * 1e6 / winstr * 256, winstr * active / (best * 1e-3));
            }
    return 0;
}
