// amc_resolve_dev.h — device side of the ordered resolve (amc_resolve.hip): the literal emulation of the reference's cell
// loops on a cluster (two particles in registers / any size from a working set / one cluster by a whole wave), the slot
// and merge-edge bookkeeping, and the validation probe.  See the header of amc_resolve.hip for the method.
#pragma once
#include "amc_grid_dev.h"

#define RS_T 512
#define RS_SORT_LDS 2048       // complex-cluster members sorted in LDS up to this many
#define RS_POOL 256            // complex-cluster working set held in LDS up to this many members
#define RS_MAX_ROUNDS 256
#define AMC_CR2_INFLATE (1.0 + 1.0e-9)

// what the resolve kernels need of amc_params (the whole structure is 416 bytes, seven cache lines of kernel arguments that
// a lone wave would fetch one dependent scalar load after the other)
struct rs_geom {
    double collision_range, argon_mass, dx, dy, dz, overlap_x, overlap_y, overlap_z;
    int nx, ny, nz, geometry;
};

struct rs_args {
    rs_geom P;
    amc_state S;
    amc_lists B;
    amc_grid G;
    amc_resolve_ws W;
    amc_out O;
    long long n;
    int allpairs;
    int defer_commit;         // leave the results in the slot arrays: the next streaming pass (or k_apply) writes them
    int apply_only;           // k_commit: only write deferred results (amc_flush)
    int force_mono;           // the host launched ONLY this kernel (it expects a small sweep): do everything here
    int plan_small;           // candidate pairs up to which the first resolve kernel does the whole sweep itself
    int *host_ncand;          // host-mapped word: candidate count of this sweep, read (lagging) by the host to pick the launch plan (saves three kernels' latency)
    int count_pp;             // this rank adds the sweep's collision count to the counters (rank 0 in multi-GPU)
    long long lo, hi;         // owned particle range: completed paths are emitted by the owner of the particle
    double inv_dx, inv_dy, inv_dz;   // 1/dx.. for floor() GUESSES only (membership is decided by the exact comparisons)
    long long *dbg;           // optional [16] phase timers (wall_clock64 ticks, 100 MHz), diagnostic only
    unsigned int sweep_epoch; // tag of the W.adj_head entries that belong to this sweep
    int wide_plan;            // k_clusters_wide ran before this kernel: start from the counters it left in W.wctl
    int wide_per;             // candidates per wave of k_clusters_wide (from the lagging candidate count)
};

// The argument block of the resolve kernels is 1.3 KB.  Read field by field from the kernarg segment it costs every wave a
// long train of dependent scalar loads (and SGPR spills) at the head of every phase — measured: most of these
// latency-bound kernels' time.  Each workgroup therefore copies the block into LDS once, with one coalesced vector
// load, and reads the fields from there.  The kernel's only formal parameter is the block (offset 0 of the segment).
#define RS_STAGE_ARGS(A)                                                                              \
    __shared__ rs_args s_args__;                                                                      \
    {                                                                                                 \
        const int *src__ = (const int *)__builtin_amdgcn_kernarg_segment_ptr();                       \
        int *dst__ = (int *)&s_args__;                                                                \
        for (int i__ = threadIdx.x; i__ < (int)(sizeof(rs_args) / 4); i__ += blockDim.x) dst__[i__] = src__[i__];   \
        __syncthreads();                                                                              \
    }                                                                                                 \
    const rs_args &A = s_args__

struct amc_ctx;
hipError_t amc_launch_clusters_wide(amc_ctx *c, const rs_args &A);      // amc_clusters.hip
int amc_clusters_wide_blocks(amc_ctx *c);

// counters of one sweep; lives in LDS while a resolve kernel runs and in W.ctl (global) between the kernels
struct rs_shared {
    int nslots, nedges, nhist, nev, dirty, changed, nhits, nfp, ovf, nclusters, ncomplex;
    int rounds, ncand, active, ok, edges_done;
    int lazy_ns;              // slots whose results are still only in the slot arrays (deferred commit)
    int nslots0;              // slots that existed (and have labels in W.sl_label) when a resolve kernel handed over
    int hist_begin;           // first history entry of the current round (older ones were validated already)
    int cur_round;
};

// working set of the multi-particle clusters (LDS pool or global fallback), indexed by sorted rank
struct rs_work {
    double *x, *y, *z, *vx, *vy, *vz, *d, *dx, *dy, *dz;
    int *tmp, *pidx, *slot;     // scratch, particle index, slot index of each member
    uint8_t *flag, *moved;
};

AMC_DEV amc_particle rs_load_particle(const amc_state &S, int p)
{
    amc_particle q;
    q.x = S.x[p]; q.y = S.y[p]; q.z = S.z[p]; q.vx = S.vx[p]; q.vy = S.vy[p]; q.vz = S.vz[p];
    q.d = S.d[p]; q.dx = S.dx[p]; q.dy = S.dy[p]; q.dz = S.dz[p]; q.flag = S.flag[p] != 0;
    return q;
}
AMC_DEV void rs_store_slot(const amc_resolve_ws &W, int s, const amc_particle &q)
{
    double *t = W.sl_state + (size_t)s * RS_SLOT_DOUBLES;          // one 96-byte record
    t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.vx; t[4] = q.vy; t[5] = q.vz;
    t[6] = q.d; t[7] = q.dx; t[8] = q.dy; t[9] = q.dz; t[10] = q.flag ? 1.0 : 0.0;
    W.sl_moved[s] = 1;
}
AMC_DEV amc_particle rs_load_slot(const amc_resolve_ws &W, int s)
{
    const double *t = W.sl_state + (size_t)s * RS_SLOT_DOUBLES;
    amc_particle q;
    q.x = t[0]; q.y = t[1]; q.z = t[2]; q.vx = t[3]; q.vy = t[4]; q.vz = t[5];
    q.d = t[6]; q.dx = t[7]; q.dy = t[8]; q.dz = t[9]; q.flag = t[10] != 0.0;
    return q;
}
// scratch state of a slot -> the particle arrays (commit)
AMC_DEV void rs_apply_slot(const amc_resolve_ws &W, const amc_state &S, int s, int p)
{
    const double *t = W.sl_state + (size_t)s * RS_SLOT_DOUBLES;
    S.x[p] = t[0]; S.y[p] = t[1]; S.z[p] = t[2]; S.vx[p] = t[3]; S.vy[p] = t[4]; S.vz[p] = t[5];
    S.d[p] = t[6]; S.dx[p] = t[7]; S.dy[p] = t[8]; S.dz[p] = t[9]; S.flag[p] = t[10] != 0.0 ? 1 : 0;
}
AMC_DEV amc_particle rs_load_work(const rs_work &K, int w)
{
    amc_particle q;
    q.x = K.x[w]; q.y = K.y[w]; q.z = K.z[w]; q.vx = K.vx[w]; q.vy = K.vy[w]; q.vz = K.vz[w];
    q.d = K.d[w]; q.dx = K.dx[w]; q.dy = K.dy[w]; q.dz = K.dz[w]; q.flag = K.flag[w] != 0;
    return q;
}
AMC_DEV void rs_store_work(const rs_work &K, int w, const amc_particle &q)
{
    K.x[w] = q.x; K.y[w] = q.y; K.z[w] = q.z; K.vx[w] = q.vx; K.vy[w] = q.vy; K.vz[w] = q.vz;
    K.d[w] = q.d; K.dx[w] = q.dx; K.dy[w] = q.dy; K.dz[w] = q.dz; K.flag[w] = q.flag ? 1 : 0;
    K.moved[w] = 1;
}

// history entry = one 32-byte record (x, y, z, slot | round << 32): one request to write it and one to read it (the
// validation probes are bound by the number of memory requests a single CU can have in flight)
AMC_DEV double4 rs_hist_make(double x, double y, double z, int slot, int gen)
{
    return make_double4(x, y, z, __longlong_as_double(((long long)gen << 32) | (unsigned int)slot));
}
AMC_DEV int rs_hist_slot(const double4 &r) { return (int)(unsigned int)(__double_as_longlong(r.w) & 0xffffffffLL); }
AMC_DEV int rs_hist_gen(const double4 &r) { return (int)(__double_as_longlong(r.w) >> 32); }

// counter += n for every lane that is here right now, with ONE atomic for the whole group (n in 0..2); returns the
// lane's own first index.  The counters live in LDS for the ordered workgroup and in global memory for the wide pair
// kernel, where a same-address atomic per hit would be a serial chain of ~12 ns each.
AMC_DEV int rs_count_add(int *counter, int n)
{
    if (__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)counter)) return atomicAdd(counter, n);   // LDS: cheap as it is
    const unsigned long long act = __ballot(1), b1 = __ballot(n >= 1), b2 = __ballot(n >= 2);
    const int lane = (int)__lane_id(), leader = __ffsll((long long)act) - 1;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    const int before = __popcll(b1 & lt) + __popcll(b2 & lt), total = __popcll(b1) + __popcll(b2);
    int base = 0;
    if (lane == leader && total) base = atomicAdd(counter, total);
    return __shfl(base, leader, 64) + before;
}

// What the wide cluster kernel (amc_clusters.hip) hands to rs_hit: its history pairs are reserved before the emulation
// starts (one counter increment per wave instead of one per hit) and every new position becomes a work item of the
// publish-and-probe phase that follows the emulation.  nullptr inside the ordered workgroup.
struct cw_item {
    double x, y, z;
    int h, own, p, pad;       // history entry, owner lane (its member list / slots), particle; pad = failed hit | emulation round << 1
};
struct rs_wide {
    // where the hit's pair of history entries comes from: the cluster's candidates bring two pairs each (4c, 4c + 2);
    // beyond that the counter (then the ordered workgroup redoes the cluster)
    const int *cnd;           // candidates of the cluster
    int ncnd;
    int *used;                // pairs taken so far
    int h_off;                // first counter-allocated entry of this sweep (4 * ncand)
    cw_item *items;           // work items of the wave (LDS)
    int *nitems;
    int cap;
    int own;
    int gen;                  // round tag of this emulation (the wide kernel re-emulates a cluster that pulled a particle in)
    int *it0;                 // where the owner's first pair of work items of this emulation went (-1: none yet)
    int *unval;               // set when a hit got entries the wave cannot publish: the ordered workgroup redoes the cluster
};

AMC_DEV void rs_store_hist(const amc_resolve_ws &W, int h, const double4 &r, bool coherent)
{
    if (coherent) {
        // read by other workgroups of the same launch: write-through stores (agent scope), one per 8 bytes
        double *d = (double *)&W.hist[h];
        __hip_atomic_store(d + 0, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 3, r.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&W.ov_next[h], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        W.hist[h] = r;
    }
}

// one hit inside an emulation: resolve p1 (= j, lower index) / p2 (= i) in registers, log events + history.
// Returns true if the particles moved.
AMC_DEV bool rs_hit(const rs_args &A, rs_shared *sh, amc_particle &p1, amc_particle &p2, int pj, int pi, int sj,
                    int si, int phase, long long cell, rs_wide *wd_in = nullptr)
{
    const amc_resolve_ws &W = A.W;
    // (the wide kernel's hooks are read ONCE, into registers: the structure itself sits in private memory — it is reached
    // through a pointer that is null in the ordered workgroup — and every field access there is a memory round trip of its own)
    const bool wide = wd_in != nullptr;
    rs_wide wd;
    if (wide) wd = *wd_in;
    // the hit's pair of history entries (h: particle j, h + 1: particle i); its events use the same two indices
    int h = -1;
    bool unval = false;
    if (wide) {
        // every candidate of the cluster brings TWO pairs of entries (4c, 4c + 2): hit q takes the first pairs in the order
        // of the candidate list, then the second ones (a cluster that pulled a particle in usually hits once more than it has
        // candidates); beyond that the counter — unpublishable here, so the ordered workgroup redoes the cluster
        const int q = (*wd.used)++;
        if (q < wd.ncnd) h = 4 * wd.cnd[q];
        else if (q < 2 * wd.ncnd) h = 4 * wd.cnd[q - wd.ncnd] + 2;
        if (h < 0) { h = wd.h_off + rs_count_add(&sh->nhist, 2); *wd.unval = 1; }
        unval = *wd.unval != 0;
    } else {
        h = rs_count_add(&sh->nhist, 2);
    }
    const bool room = h + 1 < W.max_hist;
    if (!room) sh->ovf = 1;
    const int gen = wide ? wd.gen : sh->cur_round;
    if (room) { W.ev_gen[h] = 0; W.ev_gen[h + 1] = 0; }
    auto emit = [&](int which, double tot, double px, double py, double pz) {
        if (!room) return;
        const int e = h + which;
        rs_event ev;
        ev.phase = phase; ev.i = pi; ev.j = pj; ev.which = which; ev.cell = cell; ev.slot = si; ev.pad = 0;
        ev.val[0] = tot; ev.val[1] = px; ev.val[2] = py; ev.val[3] = pz;
        W.ev[e] = ev;                   // one 64-byte record
        W.ev_gen[e] = gen;
    };
    const int fail = amc_collide(p1, p2, A.P.collision_range, A.P.argon_mass, emit);
    // counted on the slot — collisions in the low half, failed contact solves (the reference would raise
    // FloatingPointError there, Pore:11,185) in the high half — so that a cluster that is emulated again starts from zero
    // (no value needed back: the thread does not wait for the memory round trip)
    atomicAdd(&W.sl_hits[si], fail ? 0x10000 : 1);
    if (room) {
        // (a failed hit moved nothing: its pair of entries stays empty — round 0 never matches a slot's round)
        rs_store_hist(W, h, rs_hist_make(p1.x, p1.y, p1.z, sj, fail ? 0 : gen), wide);
        rs_store_hist(W, h + 1, rs_hist_make(p2.x, p2.y, p2.z, si, fail ? 0 : gen), wide);
        if (wide && !unval) {
            // the wave publishes and probes the two new positions after the emulation (amc_clusters.hip); the empty
            // pair of a failed hit keeps its place in the list (pad = 1: neither published nor probed)
            const int it = atomicAdd(wd.nitems, 2);
            if (it + 2 <= wd.cap) {
                if (*wd.it0 < 0) *wd.it0 = it;               // (the owner's later pairs follow it directly only if no other lane of
                                                            // the wave appended in between: cw_probe_overlay checks before it relies on that)
                cw_item a, b;
                a.x = p1.x; a.y = p1.y; a.z = p1.z; a.h = h; a.own = wd.own; a.p = pj; a.pad = fail | (wd.gen << 1);
                b.x = p2.x; b.y = p2.y; b.z = p2.z; b.h = h + 1; b.own = wd.own; b.p = pi; b.pad = fail | (wd.gen << 1);
                wd.items[it] = a;
                wd.items[it + 1] = b;
            } else {
                *wd.unval = 1;
            }
        }
    }
    return !fail;
}

AMC_DEV int rs_pore_cell(const rs_geom &P, double x, double y, double z, int gx, int gy, int gz)
{
    const int lx = amc_axis_cell(x, gx, P.nx, P.nx, P.dx, P.overlap_x);
    if (lx < 0) return -1;
    const int ly = amc_axis_cell(y, gy, P.ny, P.ny, P.dy, P.overlap_y);
    if (ly < 0) return -1;
    const int lz = amc_axis_cell(z, gz, P.nz / 2, 0, P.dz, P.overlap_z);
    if (lz < 0) return -1;
    return (lx * P.ny + ly) * (P.nz / 2) + lz;                                              // Pore:530 list order
}

// The (at most two) integers k with  k*d - ov < v < (k+1)*d  (Pore:527-529 with k = 2*layer+group-offset): the core
// cell of v and, if v lies in the overlap strip, the next one.  INT_MIN marks "none".
AMC_DEV void rs_axis_k(double v, double d, double inv_d, double ov, int &ka_out, int &kb_out)
{
    // (found in locals and handed out once: written through the references inside the loop, the callers' arrays stayed in
    // private memory — 230 scratch accesses in the pore instantiation of the wide cluster kernel)
    int ka = (int)0x80000000, kb = (int)0x80000000;
    const double f = floor(v * inv_d);
    if (f > -1.0e9 && f < 1.0e9) {
#pragma unroll
        for (int dk = -1; dk <= 1; dk++) {
            const int k = (int)f + dk;
            const double lo = (double)k * d - ov, hi = (double)(k + 1) * d;
            const bool in = lo < v && v < hi;
            const bool first = ka == (int)0x80000000;
            kb = (in && !first) ? k : kb;
            ka = (in && first) ? k : ka;
        }
    }
    ka_out = ka; kb_out = kb;
}
AMC_DEV void rs_pore_ks(const rs_args &A, const amc_particle &q, int *k)
{
    const rs_geom &P = A.P;
    rs_axis_k(q.x, P.dx, A.inv_dx, P.overlap_x, k[0], k[1]);
    rs_axis_k(q.y, P.dy, A.inv_dy, P.overlap_y, k[2], k[3]);
    rs_axis_k(q.z, P.dz, A.inv_dz, P.overlap_z, k[4], k[5]);
}
// layer of colour group `grp` along one axis from the cached k's (same rule as amc_axis_cell), -1 if none
AMC_DEV int rs_layer_from_k(int ka, int kb, int grp, int nlayers, int offset)
{
    for (int t = 0; t < 2; t++) {
        const int k = t ? kb : ka;
        if (k == (int)0x80000000) continue;
        const int twol = k - grp + offset;
        if (twol < 0 || (twol & 1)) continue;
        const int l = twol / 2;
        if (l < nlayers) return l;
    }
    return -1;
}
AMC_DEV int rs_pore_cell_k(const rs_geom &P, const int *k, int g)
{
    const int lx = rs_layer_from_k(k[0], k[1], g >> 2, P.nx, P.nx);
    if (lx < 0) return -1;
    const int ly = rs_layer_from_k(k[2], k[3], (g >> 1) & 1, P.ny, P.ny);
    if (ly < 0) return -1;
    const int lz = rs_layer_from_k(k[4], k[5], g & 1, P.nz / 2, 0);
    if (lz < 0) return -1;
    return (lx * P.ny + ly) * (P.nz / 2) + lz;                                              // Pore:530 list order
}

// smallest layer l >= from with  l*d - ov < v < (l+1)*d  for BOTH v1 and v2 (Cube:233), or -1
AMC_DEV int rs_next_common(double v1, double v2, double d, double inv_d, double ov, int n, int from)
{
    const double vmin = v1 < v2 ? v1 : v2, vmax = v1 < v2 ? v2 : v1;
    // the layer of vmax's core interval is the only one that can also hold a smaller coordinate; one below / above
    // are tested as well so that the floor() guess never decides membership (the comparisons do)
    const double f = floor(vmax * inv_d);
    if (!(f > -2.0 && f < 1.0e9)) return -1;
    int l0 = (int)f - 1;
    if (l0 < from) l0 = from;
    int l1 = (int)f + 1;
    if (l1 > n - 1) l1 = n - 1;
    for (int l = l0; l <= l1; l++) {
        const double lo = l * d - ov, hi = (l + 1) * d;
        if ((lo < vmin) && (vmax < hi)) return l;
    }
    return -1;
}

// ---- two-particle cluster: literal emulation with both particles in registers -----------------------------------------
// (in / out form: p1, p2 come back as the emulation left them; returns whether anything moved)
template <int GEOM>
AMC_DEV bool rs_emulate_pair_io(const rs_args &A, rs_shared *sh, amc_particle &p1, amc_particle &p2, int pj, int pi, int sj, int si,
                                rs_wide *wd = nullptr)
{
    const rs_geom &P = A.P;
    bool moved = false;
    const double cr = P.collision_range;
    if (GEOM == AMC_GEOM_CELL) {
        if (amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr)) moved |= rs_hit(A, sh, p1, p2, pj, pi, sj, si, 16, 0, wd);
    } else if (GEOM == AMC_GEOM_CUBE) {
        // Cube:231-238.  A coordinate lies in at most two overlapping layers, so the layers holding BOTH members are
        // enumerated directly (rs_next_common) instead of walking all nx*ny*nz cells — lanes of a wave would each
        // enter the nested loops at different iterations and the wave would execute the whole nest.
        // The stale masks come out of the structure: the x test is made once when an x-layer starts, the y test
        // once per (x,y)-layer, the z test per cell, each from the state at that moment (in_x_layer / in_y_layer /
        // in_z_layer); none is re-evaluated after a hit inside the layer.
        // (the overlap test depends on the positions only: it is re-evaluated after a hit, and once the pair no longer
        // overlaps no later cell can do anything — the remaining layers are not enumerated)
        bool ov = amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr);
        for (int lx = ov ? rs_next_common(p1.x, p2.x, P.dx, A.inv_dx, P.overlap_x, P.nx, 0) : -1; lx >= 0;
             lx = ov ? rs_next_common(p1.x, p2.x, P.dx, A.inv_dx, P.overlap_x, P.nx, lx + 1) : -1)
            for (int ly = rs_next_common(p1.y, p2.y, P.dy, A.inv_dy, P.overlap_y, P.ny, 0); ly >= 0;
                 ly = ov ? rs_next_common(p1.y, p2.y, P.dy, A.inv_dy, P.overlap_y, P.ny, ly + 1) : -1)
                for (int lz = rs_next_common(p1.z, p2.z, P.dz, A.inv_dz, P.overlap_z, P.nz, 0); lz >= 0;
                     lz = ov ? rs_next_common(p1.z, p2.z, P.dz, A.inv_dz, P.overlap_z, P.nz, lz + 1) : -1)
                    if (ov) {
                        moved |= rs_hit(A, sh, p1, p2, pj, pi, sj, si, 16, ((long long)lx * P.ny + ly) * P.nz + lz, wd);
                        ov = amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr);
                    }
    } else {
        // Pore:522-530.  Along one axis a coordinate belongs to at most two overlapping cells k (one of each parity);
        // they are found once per particle (rs_axis_k) and re-derived only after a hit moved the particles, instead of
        // dividing 48 times per pair.  Membership itself is decided by the reference's own comparisons.
        int k1[6], k2[6];
        rs_pore_ks(A, p1, k1);
        rs_pore_ks(A, p2, k2);
        // Lanes of a wave hold different pairs whose first shared colour group differs; with the hit inside the group
        // loop the wave would run the (large) collision path once per group.  So each lane first SEARCHES its next
        // group with a shared cell (cheap integer work), then all lanes resolve together, then the search resumes.
        // (Tried in round 3: the shared layer per axis and group bit computed up front, six evaluations instead of two cell
        // ids per group searched — slower, 7.2 -> 8.2 us per pair: the search usually ends at one of the first groups.)
        int g = 0;
        bool ov = amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr);      // unchanged until a hit moves the pair
        for (;;) {
            int hit_g = -1, hit_c = -1;
            if (ov)
                for (; g < 8; g++) {                                                        // Pore:522-524
                    const int c1 = rs_pore_cell_k(P, k1, g);
                    if (c1 >= 0 && c1 == rs_pore_cell_k(P, k2, g)) { hit_g = g; hit_c = c1; break; }
                }
            if (hit_g < 0) break;
            if (rs_hit(A, sh, p1, p2, pj, pi, sj, si, 16 + hit_g, hit_c, wd)) {
                moved = true;
                ov = amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr);
                if (ov) {                   // (the cells are only needed if the pair can hit again: it almost never can)
                    rs_pore_ks(A, p1, k1);
                    rs_pore_ks(A, p2, k2);
                }
            }
            g = hit_g + 1;
        }
    }
    if (moved) {
        rs_store_slot(A.W, sj, p1);
        rs_store_slot(A.W, si, p2);
    }
    return moved;
}
template <int GEOM>
AMC_DEV void rs_emulate_pair(const rs_args &A, rs_shared *sh, amc_particle p1, amc_particle p2, int pj, int pi, int sj, int si,
                             rs_wide *wd = nullptr)
{
    rs_emulate_pair_io<GEOM>(A, sh, p1, p2, pj, pi, sj, si, wd);
}

// The first hit of a cluster's emulation when it is already known (k_clusters_wide: a pair that was emulated, published
// nothing yet, and then pulled a third particle in).  The pulled-in particle overlapped nobody before that hit — it was in
// no candidate — and layers / colour groups are searched for OVERLAPPING pairs only, so the grown cluster's emulation
// reaches the pair's first hit in the same cell with the same operands: its result is installed instead of computed again
// (history entries, events and the hit count of that hit stay as the pair's emulation left them, same round tag).
struct rs_first_hit {
    int pj, pi;                 // the pair (j < i)
    amc_particle p1, p2;        // their state after the hit
};

template <int GEOM, int M>
AMC_DEV void rs_emulate_small(const rs_args &A, rs_shared *sh, amc_particle (&q)[M], const int (&pidx)[M],
                              const int (&slot)[M], bool (&moved)[M], rs_wide *wd, bool known, const rs_first_hit &fh);
template <int GEOM, int M>
AMC_DEV void rs_emulate_small(const rs_args &A, rs_shared *sh, amc_particle (&q)[M], const int (&pidx)[M],
                              const int (&slot)[M], bool (&moved)[M], rs_wide *wd)
{
    rs_first_hit none;
    none.pj = none.pi = -1; none.p1 = q[0]; none.p2 = q[0];
    rs_emulate_small<GEOM, M>(A, sh, q, pidx, slot, moved, wd, false, none);
}

// ---- small cluster (M = 3 or 4 members), all members in registers of ONE lane ----------------------------------------
// The same literal emulation as rs_emulate_generic below, with the member loops unrolled so that every member index is a
// compile-time constant (no LDS working set, no wave-level synchronisation): members in ascending particle index,
// pairs (c, a) with c < a in the order a = 1.., c = 0..a-1 (Pore:168-169), membership masks taken at the moments the
// reference takes them (Cube:233-238 stale in_x / in_y / in_z; Pore:527-530 at the gather of a colour group).
// What keeps it short: the overlap test of a pair depends on the positions only, so it is evaluated once per pair and
// again only for the pairs of a particle that a hit has moved; layers and colour groups are searched for OVERLAPPING
// pairs only (a cell in which no pair overlaps does nothing), and when no pair overlaps any more the rest of the sweep
// cannot do anything.
template <int M>
AMC_DEV int rs_small_next_layer(const double (&v)[M], const bool (&ok)[M], const bool (&ovp)[M][M], double d, double inv_d,
                                double ov, int n, int from)
{
    int best = -1;
#pragma unroll
    for (int a = 1; a < M; a++)
#pragma unroll
        for (int c = 0; c < a; c++) {
            if (!ovp[c][a] || !ok[a] || !ok[c]) continue;
            const int l = rs_next_common(v[a], v[c], d, inv_d, ov, n, from);
            if (l >= 0 && (best < 0 || l < best)) best = l;
        }
    return best;
}

template <int GEOM, int M>
AMC_DEV void rs_emulate_small(const rs_args &A, rs_shared *sh, amc_particle (&q)[M], const int (&pidx)[M],
                              const int (&slot)[M], bool (&moved)[M], rs_wide *wd, bool known, const rs_first_hit &fh)
{
    // (known: the first hit is fh — handed over by reference and a flag, not by a pointer that may be null: the structure
    // has to stay in registers)
    const rs_geom &P = A.P;
    const double cr = P.collision_range;
    bool ovp[M][M];                     // ovp[c][a], c < a: the pair overlaps at the current positions
    bool any = false;
#pragma unroll
    for (int a = 1; a < M; a++)
#pragma unroll
        for (int c = 0; c < a; c++) {
            ovp[c][a] = amc_overlap(q[c].x, q[c].y, q[c].z, q[a].x, q[a].y, q[a].z, cr);
            any |= ovp[c][a];
        }
    // a hit on (c0, a0): collide, then re-evaluate the pairs that contain one of the two
    auto hit = [&](int c0, int a0, amc_particle &p1, amc_particle &p2, int pj, int pi, int sj, int si, int phase, long long cell) {
        if (known) {
            known = false;
            if (pj == fh.pj && pi == fh.pi) { p1 = fh.p1; p2 = fh.p2; moved[c0] = true; moved[a0] = true; }
            else if (wd) *wd->unval = 1;            // (cannot happen, see rs_first_hit; if it did, the ordered workgroup redoes the cluster)
        } else if (rs_hit(A, sh, p1, p2, pj, pi, sj, si, phase, cell, wd)) { moved[c0] = true; moved[a0] = true; }
        any = false;
#pragma unroll
        for (int a = 1; a < M; a++)
#pragma unroll
            for (int c = 0; c < a; c++) {
                if (c == c0 || c == a0 || a == c0 || a == a0)
                    ovp[c][a] = amc_overlap(q[c].x, q[c].y, q[c].z, q[a].x, q[a].y, q[a].z, cr);
                any |= ovp[c][a];
            }
    };
    if (GEOM == AMC_GEOM_CELL) {
#pragma unroll
        for (int a = 1; a < M; a++)
#pragma unroll
            for (int c = 0; c < a; c++)
                if (ovp[c][a]) hit(c, a, q[c], q[a], pidx[c], pidx[a], slot[c], slot[a], 16, 0);
    } else if (GEOM == AMC_GEOM_CUBE) {
        bool all[M], in_x[M], in_y[M], in_z[M];
        double v[M];
#pragma unroll
        for (int a = 0; a < M; a++) all[a] = true;
        for (int lx = 0; any;) {
#pragma unroll
            for (int a = 0; a < M; a++) v[a] = q[a].x;
            lx = rs_small_next_layer<M>(v, all, ovp, P.dx, A.inv_dx, P.overlap_x, P.nx, lx);
            if (lx < 0) break;
            const double xlo = lx * P.dx - P.overlap_x, xhi = (lx + 1) * P.dx;              // Cube:233
#pragma unroll
            for (int a = 0; a < M; a++) in_x[a] = (xlo < q[a].x) && (q[a].x < xhi);
            for (int ly = 0; any;) {
#pragma unroll
                for (int a = 0; a < M; a++) v[a] = q[a].y;
                ly = rs_small_next_layer<M>(v, in_x, ovp, P.dy, A.inv_dy, P.overlap_y, P.ny, ly);
                if (ly < 0) break;
                const double ylo = ly * P.dy - P.overlap_y, yhi = (ly + 1) * P.dy;          // Cube:235
#pragma unroll
                for (int a = 0; a < M; a++) in_y[a] = in_x[a] && (ylo < q[a].y) && (q[a].y < yhi);
                for (int lz = 0; any;) {
#pragma unroll
                    for (int a = 0; a < M; a++) v[a] = q[a].z;
                    lz = rs_small_next_layer<M>(v, in_y, ovp, P.dz, A.inv_dz, P.overlap_z, P.nz, lz);
                    if (lz < 0) break;
                    const double zlo = lz * P.dz - P.overlap_z, zhi = (lz + 1) * P.dz;      // Cube:237
#pragma unroll
                    for (int a = 0; a < M; a++) in_z[a] = in_y[a] && (zlo < q[a].z) && (q[a].z < zhi);
                    const long long cell = ((long long)lx * P.ny + ly) * P.nz + lz;
#pragma unroll
                    for (int a = 1; a < M; a++)
#pragma unroll
                        for (int c = 0; c < a; c++)
                            if (in_z[a] && in_z[c] && ovp[c][a])
                                hit(c, a, q[c], q[a], pidx[c], pidx[a], slot[c], slot[a], 16, cell);
                    lz++;
                }
                ly++;
            }
            lx++;
        }
    } else {
        // Pore:522-530: the (at most two) cells k of every coordinate are found once per member (rs_axis_k) and again only
        // after a hit moved it; membership of a colour group's cell is decided at the gather of the group
        int ks[M][6];
#pragma unroll
        for (int a = 0; a < M; a++) rs_pore_ks(A, q[a], ks[a]);
        for (int g = 0; g < 8 && any; g++) {                                                 // Pore:522-524
            int cell[M];
#pragma unroll
            for (int a = 0; a < M; a++) cell[a] = rs_pore_cell_k(P, ks[a], g);              // membership at gather time
#pragma unroll
            for (int a = 1; a < M; a++)
#pragma unroll
                for (int c = 0; c < a; c++)
                    if (ovp[c][a] && cell[a] >= 0 && cell[c] == cell[a]) {
                        hit(c, a, q[c], q[a], pidx[c], pidx[a], slot[c], slot[a], 16 + g, cell[a]);
                        if (any) {          // (cells are looked at again only while some pair overlaps)
                            rs_pore_ks(A, q[c], ks[c]);
                            rs_pore_ks(A, q[a], ks[a]);
                        }
                    }
        }
    }
}

// ---- generic cluster (3+ members): literal emulation on the working set [b,e) ----------------------
AMC_DEV void rs_test_work(const rs_args &A, rs_shared *sh, const rs_work &K, int wj, int wi, int phase, long long cell,
                          rs_wide *wd = nullptr)
{
    if (!amc_overlap(K.x[wj], K.y[wj], K.z[wj], K.x[wi], K.y[wi], K.z[wi], A.P.collision_range)) return;
    amc_particle p1 = rs_load_work(K, wj), p2 = rs_load_work(K, wi);
    const int pj = K.pidx[wj], pi = K.pidx[wi];
    if (rs_hit(A, sh, p1, p2, pj, pi, K.slot[wj], K.slot[wi], phase, cell, wd)) {
        rs_store_work(K, wj, p1);
        rs_store_work(K, wi, p2);
    }
}

// smallest layer >= from that holds at least two members whose K.tmp has all bits of `need` set (0 = any member)
AMC_DEV int rs_next_layer(const rs_work &K, int b, int e, const double *v, int need, double d, double inv_d, double ov,
                          int n, int from)
{
    int best = -1;
    for (int a = b + 1; a < e; a++) {
        if ((K.tmp[a] & need) != need) continue;
        for (int c = b; c < a; c++) {
            if ((K.tmp[c] & need) != need) continue;
            const int l = rs_next_common(v[a], v[c], d, inv_d, ov, n, from);
            if (l >= 0 && (best < 0 || l < best)) best = l;
        }
    }
    return best;
}

AMC_DEV void rs_emulate_generic(const rs_args &A, rs_shared *sh, const rs_work &K, int b, int e, rs_wide *wd = nullptr)
{
    const rs_geom &P = A.P;
    if (P.geometry == AMC_GEOM_CELL) {
        for (int a = b + 1; a < e; a++)                                                     // Pore:168-169
            for (int c = b; c < a; c++) rs_test_work(A, sh, K, c, a, 16, 0, wd);
    } else if (P.geometry == AMC_GEOM_CUBE) {
        // Cube:231-238 for a cluster: only layers that hold at least two members can do anything, so the next such
        // layer is found from the member pairs (rs_next_common) instead of walking all nx*ny*nz cells.  K.tmp bit0/1/2 =
        // the in_x / in_y / in_z masks, each taken when its layer starts (they stay stale inside it, as in the reference).
        for (int lx = rs_next_layer(K, b, e, K.x, 0, P.dx, A.inv_dx, P.overlap_x, P.nx, 0); lx >= 0;
             lx = rs_next_layer(K, b, e, K.x, 0, P.dx, A.inv_dx, P.overlap_x, P.nx, lx + 1)) {
            const double xlo = lx * P.dx - P.overlap_x, xhi = (lx + 1) * P.dx;               // Cube:233
            for (int a = b; a < e; a++) K.tmp[a] = ((xlo < K.x[a]) && (K.x[a] < xhi)) ? 1 : 0;
            for (int ly = rs_next_layer(K, b, e, K.y, 1, P.dy, A.inv_dy, P.overlap_y, P.ny, 0); ly >= 0;
                 ly = rs_next_layer(K, b, e, K.y, 1, P.dy, A.inv_dy, P.overlap_y, P.ny, ly + 1)) {
                const double ylo = ly * P.dy - P.overlap_y, yhi = (ly + 1) * P.dy;           // Cube:235
                for (int a = b; a < e; a++) {
                    const int t = K.tmp[a] & 1;
                    K.tmp[a] = t | ((t && (ylo < K.y[a]) && (K.y[a] < yhi)) ? 2 : 0);
                }
                for (int lz = rs_next_layer(K, b, e, K.z, 3, P.dz, A.inv_dz, P.overlap_z, P.nz, 0); lz >= 0;
                     lz = rs_next_layer(K, b, e, K.z, 3, P.dz, A.inv_dz, P.overlap_z, P.nz, lz + 1)) {
                    const double zlo = lz * P.dz - P.overlap_z, zhi = (lz + 1) * P.dz;       // Cube:237
                    for (int a = b; a < e; a++) {
                        const int t = K.tmp[a] & 3;
                        K.tmp[a] = t | ((t == 3 && (zlo < K.z[a]) && (K.z[a] < zhi)) ? 4 : 0);
                    }
                    const long long cell = ((long long)lx * P.ny + ly) * P.nz + lz;
                    for (int a = b + 1; a < e; a++) {
                        if (K.tmp[a] != 7) continue;
                        for (int c = b; c < a; c++)
                            if (K.tmp[c] == 7) rs_test_work(A, sh, K, c, a, 16, cell, wd);
                    }
                }
            }
        }
    } else {
        for (int g = 0; g < 8; g++) {                                                        // Pore:522-524
            const int gx = g >> 2, gy = (g >> 1) & 1, gz = g & 1;
            int cnt = 0;
            for (int a = b; a < e; a++) {
                const int cell = rs_pore_cell(P, K.x[a], K.y[a], K.z[a], gx, gy, gz);
                K.tmp[a] = cell;
                cnt += cell >= 0;
            }
            if (cnt < 2) continue;
            for (int a = b + 1; a < e; a++) {
                const int ca = K.tmp[a];
                if (ca < 0) continue;
                for (int c = b; c < a; c++)
                    if (K.tmp[c] == ca) rs_test_work(A, sh, K, c, a, 16 + g, ca, wd);
            }
        }
    }
}

// bitonic sort of m (power of two) 64-bit keys; keys may live in LDS or global memory
AMC_DEV void rs_bitonic(unsigned long long *keys, int m)
{
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += RS_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// slot bookkeeping: particle of the slot (global), cluster label = lowest slot id of the cluster and cluster size
// (LDS when the sweep is small enough — the usual case — else the global work space)
struct rs_slots {
    int *label, *size;
    int cap;
};

// get (or create) the slot of particle p; creation is published by the barrier / kernel boundary that follows
AMC_DEV void rs_claim_slot(const amc_resolve_ws &W, rs_shared *sh, int cap, int p)
{
    const int old = atomicCAS(&W.slot_of[p], -1, -2);
    if (old == -1) {
        const int s = atomicAdd(&sh->nslots, 1);
        if (s < cap) {
            W.sl_meta[s] = make_int4(p, s, 0, 0);
            W.slot_of[p] = s;
        } else {
            sh->ovf = 1;
            W.slot_of[p] = -1;
        }
    }
}

// union of two clusters in a parent map over label values (M[l] = l for a root; roots are the minimum label of their set)
AMC_DEV int rs_find(const int *M, int x)
{
    while (M[x] != x) x = M[x];
    return x;
}
AMC_DEV void rs_union(int *M, int a, int b)
{
    for (;;) {
        a = rs_find(M, a);
        b = rs_find(M, b);
        if (a == b) return;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int old = atomicMin(&M[hi], lo);
        if (old == hi) return;
        a = old; b = lo;
    }
}

// merge request found by validation: particles pa, pb must be in one cluster (slot ids are filled in next round)
AMC_DEV void rs_add_edge(const amc_resolve_ws &W, rs_shared *sh, int pa, int pb)
{
    const int k = atomicAdd(&sh->nedges, 1);
    if (k < W.max_edges) { W.edge_a[k] = pa; W.edge_b[k] = pb; } else sh->ovf = 1;
    sh->dirty = 1;
}


// validation probe of history entry h: its position against every particle outside its cluster.  `cnt` are the
// sweep counters (LDS inside a resolve kernel, W.ctl in the wide validate kernel), `label` the per-slot labels.
// ---- the same emulation, run by a whole wave for ONE cluster -------------------------------------------------------------
// A cluster is emulated by a single thread of control (the order of the pair tests is the reference's), and with one lane
// doing it every load from the working set is a full LDS round trip and every instruction a wave-wide issue.  What does
// not depend on the order — finding the next layer that holds two members (one lane per member PAIR, then a wave
// minimum), taking the membership masks of all members, finding the members' pore cells — is spread over the lanes here;
// only the pair tests of a cell stay on lane 0.  Control flow is uniform; the lanes meet at wave-level fences.
#define RS_COOP_MAX 11      // members: 55 pairs fit the 64 lanes
// The lanes of ONE wave meet here after exchanging data.  A workgroup-scope fence waits for EVERY outstanding vector
// memory operation of the wave (vmcnt(0)): right where the working set may live in global memory (the ordered
// workgroup's large-sweep fallback), but a full memory round trip at every meeting point for a wave that has stores or
// value-less atomics in flight and exchanges through LDS only — LDS operations of a wave execute in order, a compiler
// barrier is all that takes.  k_clusters_wide (working set always in LDS) compiles with RS_WAVE_SYNC_LDS_ONLY: it
// states the global-memory orderings it needs itself (write-through stores + s_waitcnt before the overlay pushes).
AMC_DEV void rs_wave_sync()
{
#ifdef RS_WAVE_SYNC_LDS_ONLY
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#endif
    __builtin_amdgcn_wave_barrier();
}
AMC_DEV int rs_wave_min_nonneg(int v)       // minimum over the lanes of the values >= 0, -1 if there is none
{
    unsigned int u = v < 0 ? 0xffffffffu : (unsigned int)v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int t = (unsigned int)__shfl_xor((int)u, o, 64);
        u = t < u ? t : u;
    }
    return u == 0xffffffffu ? -1 : (int)u;
}
// lane's pair (a > c) of the cluster [b, e), or a = -1 for lanes beyond the number of pairs
AMC_DEV void rs_lane_pair(int lane, int m, int &a, int &c)
{
    a = -1; c = -1;
    int k = lane;
    for (int i = 1; i < m; i++) {
        if (k < i) { a = i; c = k; return; }
        k -= i;
    }
}
AMC_DEV int rs_next_layer_coop(const rs_work &K, int b, int pa, int pc, const double *v, int need, double d, double inv_d,
                               double ov, int n, int from)
{
    int l = -1;
    if (pa >= 0 && (K.tmp[b + pa] & need) == need && (K.tmp[b + pc] & need) == need)
        l = rs_next_common(v[b + pa], v[b + pc], d, inv_d, ov, n, from);
    return rs_wave_min_nonneg(l);
}

AMC_DEV void rs_emulate_coop(const rs_args &A, rs_shared *sh, const rs_work &K, int b, int e, rs_wide *wd = nullptr)
{
    const rs_geom &P = A.P;
    const int lane = threadIdx.x & 63, m = e - b;
    const int w = b + lane;                 // my member (lanes < m)
    int pa, pc;
    rs_lane_pair(lane, m, pa, pc);
    if (P.geometry == AMC_GEOM_CELL) {
        if (lane == 0)
            for (int a = b + 1; a < e; a++)                                                 // Pore:168-169
                for (int c = b; c < a; c++) rs_test_work(A, sh, K, c, a, 16, 0, wd);
        rs_wave_sync();
    } else if (P.geometry == AMC_GEOM_CUBE) {
        // Cube:231-238, structure as in rs_emulate_generic (stale in_x / in_y / in_z masks in K.tmp bit 0/1/2)
        for (int lx = rs_next_layer_coop(K, b, pa, pc, K.x, 0, P.dx, A.inv_dx, P.overlap_x, P.nx, 0); lx >= 0;
             lx = rs_next_layer_coop(K, b, pa, pc, K.x, 0, P.dx, A.inv_dx, P.overlap_x, P.nx, lx + 1)) {
            const double xlo = lx * P.dx - P.overlap_x, xhi = (lx + 1) * P.dx;               // Cube:233
            if (lane < m) K.tmp[w] = ((xlo < K.x[w]) && (K.x[w] < xhi)) ? 1 : 0;
            rs_wave_sync();
            for (int ly = rs_next_layer_coop(K, b, pa, pc, K.y, 1, P.dy, A.inv_dy, P.overlap_y, P.ny, 0); ly >= 0;
                 ly = rs_next_layer_coop(K, b, pa, pc, K.y, 1, P.dy, A.inv_dy, P.overlap_y, P.ny, ly + 1)) {
                const double ylo = ly * P.dy - P.overlap_y, yhi = (ly + 1) * P.dy;           // Cube:235
                if (lane < m) {
                    const int t = K.tmp[w] & 1;
                    K.tmp[w] = t | ((t && (ylo < K.y[w]) && (K.y[w] < yhi)) ? 2 : 0);
                }
                rs_wave_sync();
                for (int lz = rs_next_layer_coop(K, b, pa, pc, K.z, 3, P.dz, A.inv_dz, P.overlap_z, P.nz, 0); lz >= 0;
                     lz = rs_next_layer_coop(K, b, pa, pc, K.z, 3, P.dz, A.inv_dz, P.overlap_z, P.nz, lz + 1)) {
                    const double zlo = lz * P.dz - P.overlap_z, zhi = (lz + 1) * P.dz;       // Cube:237
                    if (lane < m) {
                        const int t = K.tmp[w] & 3;
                        K.tmp[w] = t | ((t == 3 && (zlo < K.z[w]) && (K.z[w] < zhi)) ? 4 : 0);
                    }
                    rs_wave_sync();
                    if (lane == 0) {
                        const long long cell = ((long long)lx * P.ny + ly) * P.nz + lz;
                        for (int a = b + 1; a < e; a++) {
                            if (K.tmp[a] != 7) continue;
                            for (int c = b; c < a; c++)
                                if (K.tmp[c] == 7) rs_test_work(A, sh, K, c, a, 16, cell, wd);
                        }
                    }
                    rs_wave_sync();
                }
            }
        }
    } else {
        for (int g = 0; g < 8; g++) {                                                        // Pore:522-524
            const int gx = g >> 2, gy = (g >> 1) & 1, gz = g & 1;
            int cell = -1;
            if (lane < m) {                                                                  // membership at gather time
                cell = rs_pore_cell(P, K.x[w], K.y[w], K.z[w], gx, gy, gz);
                K.tmp[w] = cell;
            }
            const int cnt = __popcll(__ballot(cell >= 0));
            rs_wave_sync();
            if (cnt < 2) continue;
            if (lane == 0)
                for (int a = b + 1; a < e; a++) {
                    const int ca = K.tmp[a];
                    if (ca < 0) continue;
                    for (int c = b; c < a; c++)
                        if (K.tmp[c] == ca) rs_test_work(A, sh, K, c, a, 16 + g, ca, wd);
                }
            rs_wave_sync();
        }
    }
}

#define RS_PF 4     // lists whose first element rs_probe prefetches
AMC_DEV void rs_probe(const rs_args &A, const amc_grid &G, rs_shared *cnt, const int *label, int ns, int cap, int h,
                      double cr2i)
{
    const amc_resolve_ws &W = A.W;
    const double4 me = W.hist[h];
    if (rs_hist_gen(me) == 0) return;               // the pair of a hit that failed (no new position)
    const int sme = rs_hist_slot(me);
    const int pme = W.sl_meta[sme].x;
    const int lme = label[sme];
    const double x = me.x, y = me.y, z = me.z;
    // only the cells overlapped by the collision_range box around the new position can hold a partner (2 to 3 on
    // average): fetch their list heads and overlay heads first, then the entries
    int c_lo[4], c_hi[4], lh[8], ovh[8];
    const int ncell = amc_grid_box_ranges(G, x, y, z, G.cr_probe, c_lo, c_hi);
    const double cr2g = G.cr2_probe;             // (list records are single precision: their test is widened, amc_internal.h)
    // list heads and overlay heads of every overlapped cell first (one memory round trip), then the entries
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lh[2 * k] = lh[2 * k + 1] = -1; ovh[2 * k] = ovh[2 * k + 1] = -1;
        if (k < ncell) {
            lh[2 * k] = amc_list_head(A.B, c_lo[k]); ovh[2 * k] = W.ov_head[c_lo[k]];
            if (c_hi[k] != c_lo[k]) { lh[2 * k + 1] = amc_list_head(A.B, c_hi[k]); ovh[2 * k + 1] = W.ov_head[c_hi[k]]; }
        }
    }
    const int nx_me = W.ov_next[h], nx_pa = W.ov_next[h ^ 1];       // to step over my own / my partner's entry without a round trip
    // The probe is a chain of dependent memory round trips, so the FIRST element of every list (grid and overlay) is
    // fetched before any is examined; longer lists (rare at ~0.25 particles per cell) continue one element at a time.
    amc_rec r0[RS_PF];
    double4 o0[RS_PF];
    int on0[RS_PF];
#pragma unroll
    for (int k = 0; k < RS_PF; k++) {
        if (lh[k] >= 0) r0[k] = A.B.rec[lh[k]];
        // history entries are allocated in pairs (the two particles of one hit): h ^ 1 is my partner's entry — same
        // cluster by construction, and for an isolated pair the only other entry nearby: skipped before any load
        while (ovh[k] >= 0 && (ovh[k] | 1) == (h | 1)) ovh[k] = (ovh[k] == h) ? nx_me : nx_pa;
        if (ovh[k] >= 0) {
            const int h2 = ovh[k];
            o0[k] = W.hist[h2]; on0[k] = W.ov_next[h2];
        }
    }
    auto grid_entry = [&](int node, const amc_rec &r) {
        const int idx = amc_node_particle(A.B, node);
        if (idx == pme) return;
        double rx, ry, rz;
        amc_rec_pos(G, r, rx, ry, rz);
        const double ax = rx - x, ay = ry - y, az = rz - z;
        if (ax * ax + ay * ay + az * az < cr2g) {
            const int so = W.slot_of[idx];
            if (so >= 0 && so < ns && label[so] == lme) return;
            if (so < 0) { rs_claim_slot(W, cnt, cap, idx); W.victim[idx] = A.sweep_epoch; }    // (a particle in no candidate)
            rs_add_edge(W, cnt, pme, idx);
        }
    };
    auto overlay_entry = [&](const double4 &o) {
        const int s2 = rs_hist_slot(o);
        if (rs_hist_gen(o) != W.sl_meta[s2].z) return;    // position of an emulation that was redone since
        if (label[s2] == lme) return;
        const double ax = o.x - x, ay = o.y - y, az = o.z - z;
        if (ax * ax + ay * ay + az * az < cr2i) rs_add_edge(W, cnt, pme, W.sl_meta[s2].x);
    };
    // pre-sweep positions of the particles binned into those cells
#pragma unroll
    for (int k = 0; k < RS_PF; k++)
        if (lh[k] >= 0) {
            grid_entry(lh[k], r0[k]);
            for (int q = amc_rec_next(r0[k]); q >= 0;) {
                const amc_rec r = A.B.rec[q];
                grid_entry(q, r);
                q = amc_rec_next(r);
            }
        }
    for (int k = RS_PF; k < 2 * ncell; k++)
        for (int q = lh[k]; q >= 0;) {
            const amc_rec r = A.B.rec[q];
            grid_entry(q, r);
            q = amc_rec_next(r);
        }
    // new positions of other clusters' members (overlay lists of the same cells)
#pragma unroll
    for (int k = 0; k < RS_PF; k++)
        if (ovh[k] >= 0) {
            overlay_entry(o0[k]);
            for (int h2 = on0[k]; h2 >= 0;) {
                if ((h2 | 1) == (h | 1)) { h2 = (h2 == h) ? nx_me : nx_pa; continue; }
                const int nx = W.ov_next[h2];
                overlay_entry(W.hist[h2]);
                h2 = nx;
            }
        }
    for (int k = RS_PF; k < 2 * ncell; k++)
        for (int h2 = ovh[k]; h2 >= 0; h2 = W.ov_next[h2]) {
            if ((h2 | 1) == (h | 1)) continue;
            overlay_entry(W.hist[h2]);
        }
}

AMC_DEV int rs_hist_cell(const rs_args &A, const amc_grid &G, int h)
{
    int cx, cy, cz;
    const double4 r = A.W.hist[h];
    amc_grid_coords(G, r.x, r.y, r.z, cx, cy, cz);
    return amc_grid_cell(G, cx, cy, cz, nullptr);
}
