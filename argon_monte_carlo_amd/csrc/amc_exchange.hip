// amc_exchange.hip — buffers of the multi-GPU exchange: the packed x|y|z|vx|vy|vz shard for the per-step
// all-gather.  Coalesced gather / scatter kernels; since between them they see the final position of every particle of
// the step exactly once (the own shard when it is packed, the others when they are unpacked), they also build the
// detection grid's per-cell lists (amc_grid_dev.h) — no separate binning pass over all n positions.
#include "amc_grid_dev.h"

// send = [6][m] (zero padded), recv = [world][6][m]
struct kin_arrays {
    double *a[6];
};

__global__ __launch_bounds__(256) void k_kin_pack(kin_arrays S, long long lo, long long hi, long long m, double *__restrict__ send,
                                                  amc_grid G, amc_lists B, amc_dev_counters *cnt)
{
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m) return;
    const bool in = lo + u < hi;
    double v[6];
#pragma unroll
    for (int e = 0; e < 6; e++) {
        v[e] = in ? S.a[e][lo + u] : 0.0;
        send[e * m + u] = v[e];
    }
    if (in) {
        bool outside = false;
        amc_list_insert(G, B, (int)(lo + u), v[0], v[1], v[2], &outside);
        if (outside) atomicOr(&cnt->flags, 8ULL);
    }
}
__global__ __launch_bounds__(256) void k_kin_unpack(kin_arrays S, long long n, int world, int rank, long long m,
                                                    const double *__restrict__ recv, amc_grid G, amc_lists B,
                                                    amc_dev_counters *cnt, int *__restrict__ slot_of)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)world * m) return;
    const int r = (int)(idx / m);
    const long long u = idx % m;
    if (r == rank) return;                                  // my own shard is already in place
    const long long base = n / world, rem = n % world;
    const long long lo = r * base + (r < rem ? r : rem), len = base + (r < rem ? 1 : 0);
    if (u >= len) return;
    const double *blk = recv + (size_t)r * 6 * (size_t)m;
    double v[6];
#pragma unroll
    for (int e = 0; e < 6; e++) {
        v[e] = blk[e * m + u];
        S.a[e][lo + u] = v[e];
    }
    // deferred commit of the previous sweep: this rank resolved the other shards' collisions too, but their results
    // arrive with the owners' blocks — only the slot is released (own particles: the streaming pass took theirs)
    if (slot_of[lo + u] >= 0) slot_of[lo + u] = -1;
    bool outside = false;
    amc_list_insert(G, B, (int)(lo + u), v[0], v[1], v[2], &outside);
    if (outside) atomicOr(&cnt->flags, 8ULL);
}
hipError_t amc_launch_kin_pack(amc_ctx *c, int world, int rank, int unpack)
{
    const long long m = c->kin_m;
    if (m <= 0) return hipSuccess;
    kin_arrays S;
    S.a[0] = c->S.x; S.a[1] = c->S.y; S.a[2] = c->S.z; S.a[3] = c->S.vx; S.a[4] = c->S.vy; S.a[5] = c->S.vz;
    amc_prof_begin(c, AMC_K_BIN_COUNT);       // (the list build is what these kernels cost)
    if (!unpack) {
        c->B.epoch++;                           // a new set of lists: this shard now, the other shards at the unpack
        c->kin_lists = true;
        hipLaunchKernelGGL(k_kin_pack, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, S, (long long)c->lo,
                           (long long)c->hi, m, c->kin_send, c->G, c->B, c->d_cnt);
    } else {
        hipLaunchKernelGGL(k_kin_unpack, dim3((unsigned)(((long long)world * m + 255) / 256)), dim3(256), 0, c->stream, S,
                           (long long)c->n, world, rank, m, c->kin_recv, c->G, c->B, c->d_cnt, c->W.slot_of);
    }
    amc_prof_end(c);
    return hipGetLastError();
}
