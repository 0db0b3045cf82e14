// Micro-benchmark: device-scope atomics on a table of list heads (the list build of k_stream): returning exchange,
// 64/32 bit, random vs index-ordered addresses, table size.  One atomic per lane, 1e6 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *tab, size_t mask, int sorted, unsigned long long *out)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t a = gid * 0x9E3779B97F4A7C15ULL;
    a = a * 6364136223846793005ULL + 1442695040888963407ULL;
    const size_t idx = sorted ? (gid & mask) : ((a >> 20) & mask);
    unsigned long long r = 0;
    if (MODE == 0) r = atomicExch(&tab[idx], (unsigned long long)gid);                   // returning, 64 bit
    if (MODE == 1) r = atomicExch((unsigned int *)&tab[idx], (unsigned int)gid);          // returning, 32 bit
    if (MODE == 2) atomicMax(&tab[idx], (unsigned long long)gid);                         // result unused
    if (MODE == 3) tab[idx] = gid;                                                        // plain store
    if (MODE == 4) r = tab[idx];                                                          // plain load
    if (r == 0x123456789ULL) out[0] = r;
}

int main()
{
    const size_t nmax = (size_t)1 << 24;
    unsigned long long *tab, *out;
    (void)hipMalloc(&tab, nmax * 8); (void)hipMemset(tab, 0, nmax * 8); (void)hipMalloc(&out, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 3907;
    const char *names[] = {"exch64 ret", "exch32 ret", "max64 noret", "store64", "load64"};
    for (int mode = 0; mode < 5; mode++)
        for (size_t n : {(size_t)1 << 20, (size_t)1 << 22, (size_t)1 << 24})
            for (int sorted : {0, 1}) {
                float best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    (void)hipEventRecord(e0);
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, sorted, out); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, sorted, out); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, sorted, out); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, sorted, out); break;
                    default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, sorted, out); break;
                    }
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("%-12s table=%3zu MB sorted=%d : %7.1f us  %.3g ops/s\n", names[mode], n * 8 >> 20, sorted, best * 1e3,
                       blocks * 256.0 / (best * 1e-3));
            }
    return 0;
}
