// amc_exchange.hip — buffers of the multi-GPU exchange.  One block per rank and step:
//     [3][m] positions of the shard (zero padded) | KB counts | KB lists of (particle, vx, vy, vz), room for capb each
// Positions change every step and travel whole.  Velocities change only in a collision or at a wall (0.1 to 3 per cent of
// the particles per step), so the owner sends just the particles whose velocity differs BITWISE from what it last
// published (shadow copy vpub, initialised from the uploaded state every rank holds): changes by walls or the energised
// re-emission, which only the owner computes, and changes by the sweep, which every rank computed but only the owner
// applied.  About 28 B per particle on the wire instead of 48.  The list is kept in KB = 16 banks (bank = workgroup & 15,
// one counter increment per workgroup): a single counter would be a serial chain of same-address atomics, ~12 ns each
// (measured: 3,000 changes per step at N = 1e5 made the kernel 13 us longer).  The banks together hold max(4096, m / 8)
// particles; a bank that overflows raises the capacity flag (bit 4) — the step is then invalid and reported as such.
// Between them the pack and unpack kernels see the final position of every particle of the step exactly once (the own
// shard when it is packed, the others when they are unpacked), so they also build the detection grid's per-cell lists
// (amc_grid_dev.h) — no separate binning pass over all n positions.
#include "amc_grid_dev.h"

#define AMC_KIN_BANKS 16

struct kin_arrays {
    double *a[6];       // x y z vx vy vz
    double *pub[3];     // velocities as published to the other ranks (own shard only)
};

__global__ __launch_bounds__(256) void k_kin_pack(kin_arrays S, long long lo, long long hi, long long m, long long capb,
                                                  double *__restrict__ send, amc_grid G, amc_lists B,
                                                  amc_dev_counters *cnt, int mode)
{
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = u < m && lo + u < hi;
    // (kept lists, amc_lists: mode 2 the full build of a cycle, 3 a step in between — the wave's pool count and where the
    // particle is filed are asked for now, with the state)
    const int wave_id = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    int keep_count0 = 0, keep_cell = -1, keep_node = -1;
    if (in && mode == 3) { keep_count0 = B.wave_count[wave_id]; keep_cell = B.cell_of[lo + u]; keep_node = B.node_of[lo + u]; }
    double pos[3] = {0.0, 0.0, 0.0};
    bool changed = false;
    double v[3];
    if (in) {
        const long long p = lo + u;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            pos[e] = S.a[e][p];
            v[e] = S.a[3 + e][p];
            changed |= __double_as_longlong(v[e]) != __double_as_longlong(S.pub[e][p]);
        }
    }
    if (u < m) {
#pragma unroll
        for (int e = 0; e < 3; e++) send[e * m + u] = pos[e];
    }
    if (in) {
        bool outside = false, overflow = false;
        if (mode == 1) {
            amc_list_insert(G, B, (int)(lo + u), pos[0], pos[1], pos[2], &outside);
        } else {
            const int nc = amc_list_keep(G, B, (int)(lo + u), pos[0], pos[1], pos[2], mode == 2, wave_id, keep_count0, keep_cell, keep_node,
                                         &outside, &overflow);
            if ((int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) B.wave_count[wave_id] = nc;
        }
        if (outside) atomicOr(&cnt->flags, 8ULL);
        if (overflow) atomicOr(&cnt->flags, 1ULL);
    }
    // slots of the changed particles: one increment of the bank's counter per workgroup
    __shared__ int s_n;
    __shared__ long long s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int mine = changed ? atomicAdd(&s_n, 1) : -1;
    __syncthreads();
    const int bank = (int)(blockIdx.x % AMC_KIN_BANKS);
    unsigned long long *counts = (unsigned long long *)(send + 3 * m);
    if (threadIdx.x == 0 && s_n > 0) s_base = (long long)atomicAdd(&counts[bank], (unsigned long long)s_n);
    __syncthreads();
    if (changed) {
        const long long slot = s_base + mine;
        if (slot < capb) {
            double *d = send + 3 * m + AMC_KIN_BANKS + 4 * ((long long)bank * capb + slot);
            d[0] = (double)(lo + u); d[1] = v[0]; d[2] = v[1]; d[3] = v[2];
#pragma unroll
            for (int e = 0; e < 3; e++) S.pub[e][lo + u] = v[e];
        } else {
            atomicOr(&cnt->flags, 16ULL);           // (not published: the bank is full)
        }
    }
}

__global__ __launch_bounds__(256) void k_kin_unpack(kin_arrays S, long long n, int world, int rank, long long m, long long capb,
                                                    const double *__restrict__ recv, amc_grid G, amc_lists B,
                                                    amc_dev_counters *cnt, int *__restrict__ slot_of,
                                                    double *__restrict__ send, int mode, int wave_base)
{
    const long long cap = capb * AMC_KIN_BANKS, per = m > cap ? m : cap;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // the all-gather that read `send` is over: its bank counters are cleared here for the next step's pack
    if (idx < AMC_KIN_BANKS) send[3 * m + idx] = 0.0;
    if (idx >= (long long)world * per) return;
    const int r = (int)(idx / per);
    const long long u = idx % per;
    if (r == rank) return;                                  // my own shard is already in place
    const double *blk = recv + (size_t)r * (size_t)(3 * m + AMC_KIN_BANKS + 4 * cap);
    const long long base = n / world, rem = n % world;
    const long long lo = r * base + (r < rem ? r : rem), len = base + (r < rem ? 1 : 0);
    if (u < len) {
        const int wave_id = wave_base + (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        int keep_count0 = 0, keep_cell = -1, keep_node = -1;
        if (mode == 3) { keep_count0 = B.wave_count[wave_id]; keep_cell = B.cell_of[lo + u]; keep_node = B.node_of[lo + u]; }
        double pos[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            pos[e] = blk[e * m + u];
            S.a[e][lo + u] = pos[e];
        }
        // deferred commit of the previous sweep: this rank resolved the other shards' collisions too, but their results
        // arrive from the owners (positions above, velocities in the lists below) — only the slot is released (own
        // particles: the streaming pass took theirs)
        if (slot_of[lo + u] >= 0) slot_of[lo + u] = -1;
        bool outside = false, overflow = false;
        if (mode == 1) {
            amc_list_insert(G, B, (int)(lo + u), pos[0], pos[1], pos[2], &outside);
        } else {
            const int nc = amc_list_keep(G, B, (int)(lo + u), pos[0], pos[1], pos[2], mode == 2, wave_id, keep_count0, keep_cell, keep_node,
                                         &outside, &overflow);
            if ((int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) B.wave_count[wave_id] = nc;
        }
        if (outside) atomicOr(&cnt->flags, 8ULL);
        if (overflow) atomicOr(&cnt->flags, 1ULL);
    }
    if (u < cap) {
        const long long bank = u / capb, k = u % capb;
        const long long count = __double_as_longlong(blk[3 * m + bank]);
        if (k == 0 && count > capb) atomicOr(&cnt->flags, 16ULL);
        if (k < count) {
            const double *d = blk + 3 * m + AMC_KIN_BANKS + 4 * u;
            const long long q = (long long)d[0];
            if (q >= lo && q < lo + len) { S.a[3][q] = d[1]; S.a[4][q] = d[2]; S.a[5][q] = d[3]; }
        }
    }
}

hipError_t amc_launch_kin_pack(amc_ctx *c, int world, int rank, int unpack)
{
    const long long m = c->kin_m, capb = c->kin_cap / AMC_KIN_BANKS;
    if (m <= 0) return hipSuccess;
    kin_arrays S;
    S.a[0] = c->S.x; S.a[1] = c->S.y; S.a[2] = c->S.z; S.a[3] = c->S.vx; S.a[4] = c->S.vy; S.a[5] = c->S.vz;
    S.pub[0] = c->kin_vpub; S.pub[1] = c->kin_vpub + c->n; S.pub[2] = c->kin_vpub + 2 * c->n;
    amc_prof_begin(c, AMC_K_BIN_COUNT);       // (the list build is what these kernels cost)
    amc_lists Bm = c->B;
    Bm.wave_count = c->mg_wave_count;       // (kept lists: the exchange kernels' own pools)
    if (!unpack) {
        // a new set of lists — this shard now, the other shards at the unpack — or, with kept lists (pore), a step of a cycle
        if (c->mg_keep) {
            if (c->lists_owner != 2) c->lists_age = -1;
            c->lists_owner = 2;
            if (c->lists_age < 0 || c->lists_age + 1 >= c->keep_K) { c->kin_mode = 2; c->B.epoch++; c->lists_age = 0; }
            else { c->kin_mode = 3; c->lists_age++; }
        } else {
            c->kin_mode = 1; c->lists_age = -1; c->B.epoch++;
        }
        Bm = c->B;
        Bm.wave_count = c->mg_wave_count;
        c->kin_lists = true;
        if (!c->kin_counts_clear) {             // (normally the previous step's unpack kernel has cleared the banks' counters)
            hipError_t e = hipMemsetAsync(c->kin_send + 3 * m, 0, sizeof(double) * AMC_KIN_BANKS, c->stream);
            if (e != hipSuccess) return e;
        }
        c->kin_counts_clear = false;
        AMC_LAUNCH(c, k_kin_pack, dim3((unsigned)((m + 255) / 256)), dim3(256), S, (long long)c->lo,
                           (long long)c->hi, m, capb, c->kin_send, c->G, Bm, c->d_cnt, c->kin_mode);
    } else {
        const long long per = m > c->kin_cap ? m : c->kin_cap;
        AMC_LAUNCH(c, k_kin_unpack, dim3((unsigned)(((long long)world * per + 255) / 256)), dim3(256), S,
                           (long long)c->n, world, rank, m, capb, c->kin_recv, c->G, Bm, c->d_cnt, c->W.slot_of, c->kin_send, c->kin_mode,
                           c->mg_waves_pack);
        c->kin_counts_clear = true;
    }
    amc_prof_end(c);
    return hipGetLastError();
}

int amc_kin_banks(void) { return AMC_KIN_BANKS; }
