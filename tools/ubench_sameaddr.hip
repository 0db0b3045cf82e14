// Micro-benchmark: returning atomicAdd on ONE address from one lane of each of W waves (the candidate counter of the
// detect kernel), against the same number of atomics spread over 64 addresses.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(unsigned int *ctr, int spread, int waves, unsigned int *out)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = gid >> 6;
    if ((gid & 63) != 0 || wave >= waves) return;
    const unsigned int r = atomicAdd(&ctr[spread ? (wave & 63) * 32 : 0], 1u);
    if (r == 0xffffffffu) out[0] = r;
}

int main()
{
    unsigned int *ctr, *out;
    (void)hipMalloc(&ctr, 64 * 32 * 4); (void)hipMemset(ctr, 0, 64 * 32 * 4); (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int waves : {500, 2000, 4000, 8000, 15625})
        for (int spread : {0, 1}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(3907), dim3(256), 0, 0, ctr, spread, waves, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("waves=%5d %s : %7.1f us  (%.1f ns per atomic)\n", waves, spread ? "64 addresses" : "one address ", best * 1e3, best * 1e6 / waves);
        }
    return 0;
}
