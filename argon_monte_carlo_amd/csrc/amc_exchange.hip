// amc_exchange.hip — buffers of the multi-GPU exchange: the packed x|y|z|vx|vy|vz shard for the per-step
// all-gather.  Plain coalesced gather / scatter kernels.
#include "amc_internal.h"

// send = [6][m] (zero padded), recv = [world][6][m]
struct kin_arrays {
    double *a[6];
};

__global__ __launch_bounds__(256) void k_kin_pack(kin_arrays S, long long lo, long long hi, long long m, double *__restrict__ send)
{
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m) return;
    const bool in = lo + u < hi;
#pragma unroll
    for (int e = 0; e < 6; e++) send[e * m + u] = in ? S.a[e][lo + u] : 0.0;
}
__global__ __launch_bounds__(256) void k_kin_unpack(kin_arrays S, long long n, int world, int rank, long long m,
                                                    const double *__restrict__ recv)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)world * m) return;
    const int r = (int)(idx / m);
    const long long u = idx % m;
    if (r == rank) return;                                  // my own shard is already in place
    const long long base = n / world, rem = n % world;
    const long long lo = r * base + (r < rem ? r : rem), cnt = base + (r < rem ? 1 : 0);
    if (u >= cnt) return;
    const double *blk = recv + (size_t)r * 6 * (size_t)m;
#pragma unroll
    for (int e = 0; e < 6; e++) S.a[e][lo + u] = blk[e * m + u];
}
hipError_t amc_launch_kin_pack(amc_ctx *c, int world, int rank, int unpack)
{
    const long long m = c->kin_m;
    if (m <= 0) return hipSuccess;
    kin_arrays S;
    S.a[0] = c->S.x; S.a[1] = c->S.y; S.a[2] = c->S.z; S.a[3] = c->S.vx; S.a[4] = c->S.vy; S.a[5] = c->S.vz;
    if (!unpack)
        hipLaunchKernelGGL(k_kin_pack, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, S, (long long)c->lo,
                           (long long)c->hi, m, c->kin_send);
    else
        hipLaunchKernelGGL(k_kin_unpack, dim3((unsigned)(((long long)world * m + 255) / 256)), dim3(256), 0, c->stream, S,
                           (long long)c->n, world, rank, m, c->kin_recv);
    return hipGetLastError();
}
