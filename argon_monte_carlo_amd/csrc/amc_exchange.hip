// amc_exchange.hip — buffers of the multi-GPU exchanges: the packed x|y|z shard for the per-step position
// all-gather and the state rows of candidate particles for the int64-SUM all-reduce.  Plain gather / scatter kernels.
#include "amc_internal.h"

// exchange table [11][n]: rows of the particles this rank owns, zero bits elsewhere (the all-reduce is an integer SUM of the int64 view: exact)
__global__ __launch_bounds__(256) void k_pack_rows(amc_state S, const int *list, int n, long long lo, long long hi, double *table)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const int p = list[u];
    const bool own = p >= lo && p < hi;
    const double v[11] = {S.x[p], S.y[p], S.z[p], S.vx[p], S.vy[p], S.vz[p], S.d[p], S.dx[p], S.dy[p], S.dz[p],
                          S.flag[p] ? 1.0 : 0.0};
    for (int e = 0; e < 11; e++) table[(size_t)e * n + u] = own ? v[e] : 0.0;
}
__global__ __launch_bounds__(256) void k_unpack_rows(amc_state S, const int *list, int n, long long lo, long long hi, const double *table)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const int p = list[u];
    if (p >= lo && p < hi) return;          // the owner's copy is authoritative
    S.x[p] = table[(size_t)0 * n + u]; S.y[p] = table[(size_t)1 * n + u]; S.z[p] = table[(size_t)2 * n + u];
    S.vx[p] = table[(size_t)3 * n + u]; S.vy[p] = table[(size_t)4 * n + u]; S.vz[p] = table[(size_t)5 * n + u];
    S.d[p] = table[(size_t)6 * n + u]; S.dx[p] = table[(size_t)7 * n + u]; S.dy[p] = table[(size_t)8 * n + u];
    S.dz[p] = table[(size_t)9 * n + u];
    S.flag[p] = table[(size_t)10 * n + u] != 0.0;
}

// ---- packed position exchange (one all-gather per step): send = [3][m], recv = [world][3][m] ---------------------------
__global__ __launch_bounds__(256) void k_pos_pack(const double *__restrict__ x, const double *__restrict__ y,
                                                  const double *__restrict__ z, long long lo, long long hi, long long m,
                                                  double *__restrict__ send)
{
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m) return;
    const bool in = lo + u < hi;
    send[u] = in ? x[lo + u] : 0.0;
    send[m + u] = in ? y[lo + u] : 0.0;
    send[2 * m + u] = in ? z[lo + u] : 0.0;
}
__global__ __launch_bounds__(256) void k_pos_unpack(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                                    long long n, int world, int rank, long long m,
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
    const double *blk = recv + (size_t)r * 3 * (size_t)m;
    x[lo + u] = blk[u];
    y[lo + u] = blk[m + u];
    z[lo + u] = blk[2 * m + u];
}
hipError_t amc_launch_pos_pack(amc_ctx *c, int world, int rank, int unpack)
{
    const long long m = c->pos_m;
    if (m <= 0) return hipSuccess;
    if (!unpack)
        hipLaunchKernelGGL(k_pos_pack, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, c->S.x, c->S.y, c->S.z,
                           (long long)c->lo, (long long)c->hi, m, c->pos_send);
    else
        hipLaunchKernelGGL(k_pos_unpack, dim3((unsigned)(((long long)world * m + 255) / 256)), dim3(256), 0, c->stream, c->S.x,
                           c->S.y, c->S.z, (long long)c->n, world, rank, m, c->pos_recv);
    return hipGetLastError();
}

hipError_t amc_launch_pack(amc_ctx *c, const int *d_list, int n, double *table, int unpack)
{
    if (n <= 0) return hipSuccess;
    if (unpack) hipLaunchKernelGGL(k_unpack_rows, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, d_list, n, c->lo, c->hi, table);
    else hipLaunchKernelGGL(k_pack_rows, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, d_list, n, c->lo, c->hi, table);
    return hipGetLastError();
}
