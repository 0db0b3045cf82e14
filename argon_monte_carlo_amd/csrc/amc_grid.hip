// amc_grid.hip — candidate detection for the p-p sweep (the parallel half of Pore:520-549 / Cube:231-336).
//
// The reference finds colliding pairs by an O(n^2) loop inside each ~21 nm cell.  Every pair closer than
// collision_range shares at least one reference cell, so ANY exact close-pair detector produces a
// superset of the pairs the reference can hit; the ordered resolve (amc_resolve.hip) then re-tests each pair with
// the reference's own cell membership and loop order.  Two detectors:
//
//   binned   : per-cell particle lists over cells of edge h ~ 0.63 x mean spacing (>= 2 collision_range), built with
//              one epoch-tagged 64-bit atomic exchange per particle (inside k_stream) - no counters to clear, no
//              scan, no scatter; then each particle walks its own list and those of the lower-numbered cells its
//              collision_range box overlaps.  O(N) work, HBM/L2-bound.  Algorithmic traffic 24 B/particle (positions read once); the
//              32-byte records and the list heads are implementation overhead.
//   all-pairs: LDS-tiled j-block (256 particles = 6 KB) against 256 i-particles in registers, upper triangle of
//              tiles only — the kernel the reference's pairwise_particles_in_cell maps to directly; fp64-VALU-bound
//              (9 flop per pair), used for single cells (amc_pairwise_cell), small N and cross-validation.
//
// A pair is emitted when d^2 < collision_range^2 * (1 + 1e-9): a superset of the reference's
// `sqrt(d^2) < collision_range` test, which the resolve re-evaluates exactly.
#include <stdlib.h>

#include "amc_grid_dev.h"

#define AMC_CR2_INFLATE (1.0 + 1.0e-9)

// ---- binning: per-cell particle lists ------------------------------------------------------------------------------------
// stand-alone form (the step driver builds the lists inside k_stream, the multi-GPU path inside its pack / unpack
// kernels; this one serves the stage API)
__global__ __launch_bounds__(256) void k_bin_lists(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, long long n, amc_grid G, amc_lists B,
                                                   amc_dev_counters *cnt, amc_ovl V)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (V.skip_epoch && (unsigned int)(V.adj_head[p] >> 32) == V.skip_epoch) return;      // (overlapped run: filed by the fix-up kernel)
    bool outside = false;
    amc_list_insert(G, B, (int)p, x[p], y[p], z[p], &outside);
    if (outside) atomicOr(&cnt->flags, 8ULL);
}

// ---- candidate bookkeeping --------------------------------------------------------------------------------------------------
// A candidate is stored as (i, j) with i > j plus a copy of both particles' state in a SoA table (cst[e][k], e = 0..10
// particle j, 11..21 particle i): the single-workgroup resolve kernel then reads coalesced rows instead of issuing
// 22 scattered loads per pair from one CU.  Returns the candidate's slot (or -1 on overflow).
// The candidate graph: every pushed pair is linked into the candidate lists of both its particles with one 64-bit
// exchange each (head tagged with the sweep's epoch, so the table is never cleared).  The wide cluster kernel
// (amc_clusters.hip) walks these lists: a pair whose two particles appear in no other candidate is emulated by one lane,
// a larger connected component by the wave of its lowest candidate.
struct amc_adj {
    unsigned long long *head;   // [n]   nullptr: the graph is not needed (all-pairs mode)
    int4 *rec;                  // [max_cand] (i, j, next in i's list, next in j's list)
    int4 *sd;                   // [max_cand] (slot of i, slot of j, done, -): reset here
    // candidate k brings its own two slots (2k, 2k + 1) and two pairs of history / event entries (4k .. 4k + 3): whoever
    // emulates the candidate's cluster uses them without any allocation; here they are marked empty
    int4 *sl_meta;
    int *sl_hits, *ev_gen;
    unsigned int epoch;
    unsigned long long *mark;
};

AMC_DEV int amc_push_candidate(int a, int b, int max_cand, amc_dev_counters *cnt, const amc_adj &D)
{
    const unsigned int k = atomicAdd(&cnt->cand_count, 1u);
    if (k < (unsigned)max_cand) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        int4 r;
        r.x = hi; r.y = lo; r.z = -1; r.w = -1;
        if (D.head) {
            const unsigned long long mine = ((unsigned long long)D.epoch << 32) | (unsigned long long)k;
            const unsigned long long oi = atomicExch(&D.head[hi], mine), oj = atomicExch(&D.head[lo], mine);
            r.z = ((unsigned int)(oi >> 32) == D.epoch) ? (int)(unsigned int)(oi & 0xffffffffULL) : -1;
            r.w = ((unsigned int)(oj >> 32) == D.epoch) ? (int)(unsigned int)(oj & 0xffffffffULL) : -1;
            // the candidates my exchanges displaced are no longer alone on their particle: told in a candidate-indexed word,
            // so that the lane of an ISOLATED pair (99 %) never has to read the particle-indexed heads (a sparsely touched
            // N-sized array: its translation misses made that one load 4.3 us of the wide kernel's 12 us chain)
            // The word also names the displacing candidate (its SUCCESSOR in that particle's list), so that the owner of a
            // two-candidate chain reaches the second candidate in one hop; a candidate displaced on both particles only says so.
            auto mark = [&](int kd) {
                const unsigned long long old = atomicExch(&D.mark[kd], mine);
                if ((unsigned int)(old >> 32) == D.epoch) D.mark[kd] = ((unsigned long long)D.epoch << 32) | AMC_MARK_MULTI;
            };
            if (r.z >= 0) mark(r.z);
            if (r.w >= 0) mark(r.w);
        }
        D.rec[k] = r;
        D.sd[k] = make_int4(-1, -1, 0, 0);      // (slot of i, slot of j, done by the wide kernel, -)
        if (D.head) {
            D.sl_meta[2 * k] = make_int4(-1, 2 * (int)k, 0, 0);
            D.sl_meta[2 * k + 1] = make_int4(-1, 2 * (int)k + 1, 0, 0);
            atomicAnd(&D.sl_hits[2 * k], 0);    // (atomics, like the increments: no value needed back)
            atomicAnd(&D.sl_hits[2 * k + 1], 0);
            *(int4 *)&D.ev_gen[4 * k] = make_int4(0, 0, 0, 0);      // (its two pairs of history / event entries: 4k .. 4k + 3)
        }
        return (int)k;
    }
    atomicOr(&cnt->flags, 1ULL);
    return -1;
}

// ---- binned detection: one thread per particle ----------------------------------------------------------------------------
// Each particle walks its own cell's list (the part inserted before it) and the lists of the lower-numbered cells its
// +-collision_range box overlaps; every close pair is met exactly once and stored as (larger index, smaller index).
// Measured on MI355X (tools/ubench_vmem.hip): random 8..32-byte loads over a >L2 footprint run at ~5.5e10 requests/s
// whatever their width, index-ordered ones at ~2e11/s — the kernel's cost is its number of random requests
// (heads + list elements, ~1.3 per particle at 0.25 particles per cell), not its bytes.
// One list NODE: its own cell's list (the part inserted before it) and the lists of the lower-numbered cells its box overlaps.
AMC_DEV void amc_detect_node(const amc_grid &G, const amc_lists &B, int node, double cr2i, double cr_probe, int max_cand,
                             amc_dev_counters *cnt, const amc_adj &D)
{
    amc_rec me_r = B.rec[node];
    // (kept lists: where the particle's live node is — asked for together with the record, so that a moved particle costs its
    // wave one more round trip, not two; at 6 % movers nearly every wave has one)
    const int live = (B.node_of && node < B.n) ? B.node_of[node] : node;
    if (me_r.x != me_r.x) {
        // a particle's own node after the particle was filed again under another one.  Kept lists: THIS thread walks for the
        // live node (one node per thread whatever moved; blocks of their own for the extra nodes cost what blocks of
        // particles cost, whether they find work or not — 2,048 of them were +21 us).  An overlapped run (no node_of): the
        // extra nodes have blocks of their own.
        if (live == node) return;
        node = live;
        me_r = B.rec[node];
        if (me_r.x != me_r.x) return;
    }
    const int me_p = amc_node_particle(B, node);
    double3 me;
    amc_rec_pos(G, me_r, me.x, me.y, me.z);
    auto found_pair = [&](int q) { amc_push_candidate(me_p, amc_node_particle(B, q), max_cand, cnt, D); };
    // Nine list cursors per particle — slot 0: my own cell, only the particles inserted BEFORE me (my `next` chain;
    // every same-cell pair is thereby met exactly once, by the later-inserted particle, and the head is not needed);
    // slots 1..8: the other cells my box overlaps, but only those with a SMALLER cell id: two particles closer than
    // collision_range lie in each other's box, so a cross-cell pair is met exactly once, from the larger cell.  The kernel is bound by the LATENCY of dependent loads (head -> record -> next record),
    // so all cursors advance together: every round issues the loads of all live cursors before using any of them.
    int q[9];
    q[0] = amc_rec_next(me_r);
    {
        int ocx, ocy, ocz;
        amc_grid_coords(G, me.x, me.y, me.z, ocx, ocy, ocz);
        const int c_own = amc_grid_cell(G, ocx, ocy, ocz, nullptr);
        int c_lo[4], c_hi[4];
        const int nc = amc_grid_box_ranges(G, me.x, me.y, me.z, cr_probe, c_lo, c_hi);
        int cell[8];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int c = c_lo[k < nc ? k : 0] + t;
                cell[2 * k + t] = (k < nc && c <= c_hi[k < nc ? k : 0] && c < c_own) ? c : -1;
            }
        unsigned long long hv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) hv[e] = (cell[e] >= 0) ? B.head[cell[e]] : 0ULL;
#pragma unroll
        for (int e = 0; e < 8; e++)
            q[1 + e] = (cell[e] >= 0 && (unsigned int)(hv[e] >> 32) == B.epoch) ? (int)(unsigned int)(hv[e] & 0xffffffffULL) : -1;
    }
    for (;;) {
        bool live = false;
#pragma unroll
        for (int e = 0; e < 9; e++) live |= q[e] >= 0;
        if (!live) break;
        amc_rec o[9];
#pragma unroll
        for (int e = 0; e < 9; e++)
            if (q[e] >= 0) o[e] = B.rec[q[e]];
#pragma unroll
        for (int e = 0; e < 9; e++)
            if (q[e] >= 0) {
                double ox, oy, oz;
                amc_rec_pos(G, o[e], ox, oy, oz);
                const double ex = ox - me.x, ey = oy - me.y, ez = oz - me.z;
                if (ex * ex + ey * ey + ez * ez < cr2i) found_pair(q[e]);      // (false for a record with a NaN position)
                q[e] = amc_rec_next(o[e]);
            }
    }
}

// One thread per particle; in an overlapped run (amc_stream.hip) a few more blocks take the extra nodes (amc_lists), whose
// number only the device knows.
#ifndef AMC_DETECT_MINW
#define AMC_DETECT_MINW 4       // waves per SIMD the register allocation has to leave room for (128 registers): at 129 the
                                // kernel ran with three and took 22 % longer in the pore at N = 1e6.  Measured beside it: 5 and
                                // 6 waves bought with spills (96 / 80 registers) 36.0 / 42.4 us against 35.9; the candidate push
                                // as a real call instead of inlined (116 registers, but the call's frame) 48 us
#endif
__global__ __launch_bounds__(256, AMC_DETECT_MINW) void k_detect_lists(amc_grid G, amc_lists B, long long n, double cr2i, double cr_probe,
                                                      int max_cand, amc_dev_counters *cnt, amc_adj D, const int *extra_count,
                                                      int max_extra)
{
    const long long nb = (n + blockDim.x - 1) / blockDim.x;
    if ((long long)blockIdx.x < nb) {
        const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (p < n) amc_detect_node(G, B, (int)p, cr2i, cr_probe, max_cand, cnt, D);
        return;
    }
    if (!extra_count) return;
    // (an overlapped run: a few more blocks for the particles its fix-up kernel filed again under extra nodes)
    int ne = *extra_count;
    if (ne > max_extra) ne = max_extra;
    const int stride = (int)((gridDim.x - nb) * blockDim.x);
    for (int e = (int)((blockIdx.x - nb) * blockDim.x + threadIdx.x); e < ne; e += stride)
        amc_detect_node(G, B, (int)n + e, cr2i, cr_probe, max_cand, cnt, D);
}

// ---- detection sharded by index (multi-GPU) ------------------------------------------------------------------
// A rank examines only ITS particles, but against everybody: particle p walks the WHOLE list of every cell its box
// overlaps, its own cell included, and keeps the pairs whose partner has the lower index — so a pair is found exactly once
// in the whole job, by the rank that owns its higher index.  The pairs go into the rank's block of a second, small
// all-gather ([0] = count, then (i, j) as two 32-bit integers each); k_ingest_candidates then builds the candidate graph
// from the blocks of all ranks, identically everywhere, and the (replicated) resolve proceeds as on one GPU.
__global__ __launch_bounds__(256) void k_detect_own(amc_grid G, amc_lists B, long long lo, long long hi, double cr2i,
                                                    double cr_probe, int *__restrict__ out, int cap, amc_dev_counters *cnt)
{
    const long long p = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hi) return;
    amc_rec me_r = B.rec[p];
    // (kept lists: a particle that has moved is walked for through its live node, asked for together with its own record)
    const int live = B.node_of ? B.node_of[p] : (int)p;
    if (me_r.x != me_r.x) {
        if (live == (int)p) return;
        me_r = B.rec[live];
        if (me_r.x != me_r.x) return;
    }
    double3 me;
    amc_rec_pos(G, me_r, me.x, me.y, me.z);
    int c_lo[4], c_hi[4];
    const int nc = amc_grid_box_ranges(G, me.x, me.y, me.z, cr_probe, c_lo, c_hi);
    int q[8];
    unsigned long long hv[8];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = c_lo[k < nc ? k : 0] + t;
            const bool on = k < nc && c <= c_hi[k < nc ? k : 0];
            hv[2 * k + t] = on ? B.head[c] : 0ULL;
            q[2 * k + t] = on ? 0 : -1;
        }
#pragma unroll
    for (int e = 0; e < 8; e++)
        q[e] = (q[e] == 0 && (unsigned int)(hv[e] >> 32) == B.epoch) ? (int)(unsigned int)(hv[e] & 0xffffffffULL) : -1;
    for (;;) {
        bool live = false;
#pragma unroll
        for (int e = 0; e < 8; e++) live |= q[e] >= 0;
        if (!live) break;
        amc_rec o[8];
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (q[e] >= 0) o[e] = B.rec[q[e]];
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (q[e] >= 0) {
                double ox, oy, oz;
                amc_rec_pos(G, o[e], ox, oy, oz);
                const double ex = ox - me.x, ey = oy - me.y, ez = oz - me.z;
                // (a list entry is a NODE: the partner is the particle it stands for; my own live node is no partner)
                const int pq = amc_node_particle(B, q[e]);
                if (pq < (int)p && ex * ex + ey * ey + ez * ez < cr2i) {
                    const int k = atomicAdd(&out[0], 1);
                    if (k < cap) { out[2 + 2 * k] = (int)p; out[3 + 2 * k] = pq; }
                    else atomicOr(&cnt->flags, 1ULL);       // (candidate overflow: the step reports AMC_ERR_CAPACITY)
                }
                q[e] = amc_rec_next(o[e]);
            }
    }
}

__global__ __launch_bounds__(256) void k_ingest_candidates(const int *__restrict__ blocks, int world, int block_ints, int cap,
                                                           int max_cand, amc_dev_counters *cnt, amc_adj D, int *__restrict__ own)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    for (int r = 0; r < world; r++) {
        const int *b = blocks + (size_t)r * block_ints;
        int nr = b[0];
        if (nr > cap) nr = cap;
        for (int k = gtid; k < nr; k += gstride) amc_push_candidate(b[2 + 2 * k], b[3 + 2 * k], max_cand, cnt, D);
    }
    if (gtid == 0) own[0] = 0;          // (this rank's block has been gathered: its counter is cleared for the next step)
}

// ---- all-pairs detection: LDS tile of 256 j-particles against 256 i-particles in registers ---------------------------
#define AP_T 256
__global__ __launch_bounds__(AP_T) void k_detect_allpairs(const double *__restrict__ x, const double *__restrict__ y,
                                                          const double *__restrict__ z, int n, int ntiles, double cr2i,
                                                          int max_cand, amc_dev_counters *cnt, amc_adj D)
{
    // blockIdx.x enumerates the lower triangle of tile pairs: (bi, bj) with bj <= bi
    int bi = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long long)(bi + 1) * (bi + 2) / 2 <= (long long)blockIdx.x) bi++;
    while ((long long)bi * (bi + 1) / 2 > (long long)blockIdx.x) bi--;
    const int bj = (int)(blockIdx.x - (long long)bi * (bi + 1) / 2);
    if (bi >= ntiles) return;
    __shared__ double tx[AP_T], ty[AP_T], tz[AP_T];
    const int j0 = bj * AP_T;
    const int jj = j0 + threadIdx.x;
    tx[threadIdx.x] = (jj < n) ? x[jj] : 0.0;
    ty[threadIdx.x] = (jj < n) ? y[jj] : 0.0;
    tz[threadIdx.x] = (jj < n) ? z[jj] : 0.0;
    __syncthreads();
    const int i = bi * AP_T + threadIdx.x;
    if (i >= n) return;
    const double xi = x[i], yi = y[i], zi = z[i];
    int jmax = n - j0;
    if (jmax > AP_T) jmax = AP_T;
    if (bi == bj && jmax > (int)threadIdx.x) jmax = threadIdx.x;   // j < i inside the diagonal tile
    // 9 flop per pair (3 differences, 3 products, 2 sums, 1 comparison — SURVEY 8d); the products and sums are issued as
    // one multiply and two explicit fused multiply-adds (the test only has to be a SUPERSET of the reference's
    // sqrt(d^2) < collision_range, and the 1e-9 inflation of cr^2 is 10^7 times the rounding difference)
#pragma unroll 4
    for (int k = 0; k < jmax; k++) {
        const double ex = tx[k] - xi, ey = ty[k] - yi, ez = tz[k] - zi;
        const double d2 = fma(ez, ez, fma(ey, ey, ex * ex));
        if (d2 < cr2i) amc_push_candidate(i, j0 + k, max_cand, cnt, D);
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------
hipError_t amc_launch_bin(amc_ctx *c)
{
    if (c->allpairs || c->n <= 0) return hipSuccess;
    const long long n = c->n;
    c->B.epoch++;
    c->lists_age = -1;          // (kept lists: these are not the streaming pass's own)
    amc_prof_begin(c, AMC_K_BIN_COUNT);
    amc_ovl V;
    V.adj_head = nullptr; V.skip_epoch = 0;
    AMC_LAUNCH(c, k_bin_lists, dim3((unsigned)((n + 255) / 256)), dim3(256), c->S.x, c->S.y, c->S.z, n,
                       c->G, c->B, c->d_cnt, V);
    amc_prof_end(c);
    return hipGetLastError();
}

// overlapped run, experiment AMC_OVERLAP_SPLIT=1: the pass's list build as a kernel of its own behind the streaming part
hipError_t amc_launch_bin_ovl(amc_ctx *c, int to, unsigned int skip_epoch, hipStream_t stream)
{
    const long long n = c->n;
    amc_ovl V;
    V.adj_head = c->W.adj_head; V.skip_epoch = skip_epoch;
    amc_prof_begin(c, AMC_K_BIN_COUNT);
    AMC_LAUNCH_ON(c, stream, k_bin_lists, dim3((unsigned)((n + 255) / 256)), dim3(256), c->S_buf[to].x, c->S_buf[to].y, c->S_buf[to].z, n,
                  c->G, c->B_buf[to], c->d_cnt, V);
    amc_prof_end(c);
    return hipGetLastError();
}

// The same detector for large N: square tiles of AP2_T = 1024 particles, every thread keeps AP2_R = 4 particles i in
// registers and walks the j-tile in LDS — a j read (four broadcast LDS loads) serves four pairs, the hit test only
// accumulates a flag, and the (rare) chunk that raised it is walked again to push its pairs.
// Arithmetic: d^2 = |ri|^2 + |rj|^2 - 2 ri.rj with the squared norms formed once per particle, so a pair costs three fused
// multiply-adds and one comparison, t = |rj|^2 - 2 ri.rj < cr^2 - |ri|^2, instead of the 3 + 3 + 2 + 1 operations of the
// direct form SURVEY 8d counts (9 flop per pair: the figure the roofline line is quoted in).  The expanded form cancels:
// with coordinates taken relative to the grid origin its error is below 13 u R^2 (u = 2^-53, R^2 = squared diagonal of the
// grid's extent), so the threshold is raised by 64 u R^2 — 7e-7 of cr^2 in the pore, 2e-9 in the cube at N = 1e5.  The
// test only has to be a SUPERSET of the reference's sqrt(d^2) < collision_range: every candidate is re-tested exactly.
#define AP2_T 1024
#define AP2_R 4             // (8 per thread on 128-thread blocks measured 47 instead of 62 TFLOP/s)
#define AP2_THREADS (AP2_T / AP2_R)
#define AP2_CHUNK 8
__global__ __launch_bounds__(AP2_THREADS) void k_detect_allpairs_tiled(const double *__restrict__ x, const double *__restrict__ y,
                                                               const double *__restrict__ z, int n, int ntiles, double thr,
                                                               double ox, double oy, double oz,
                                                               int max_cand, amc_dev_counters *cnt, amc_adj D)
{
    int bi = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((long long)(bi + 1) * (bi + 2) / 2 <= (long long)blockIdx.x) bi++;
    while ((long long)bi * (bi + 1) / 2 > (long long)blockIdx.x) bi--;
    const int bj = (int)(blockIdx.x - (long long)bi * (bi + 1) / 2);
    if (bi >= ntiles) return;
    __shared__ double tx[AP2_T], ty[AP2_T], tz[AP2_T], tn[AP2_T];
    const int j0 = bj * AP2_T;
    for (int t = threadIdx.x; t < AP2_T; t += AP2_THREADS) {
        const int jj = j0 + t;
        const bool in = jj < n;
        const double X = in ? x[jj] - ox : 0.0, Y = in ? y[jj] - oy : 0.0, Z = in ? z[jj] - oz : 0.0;
        tx[t] = X; ty[t] = Y; tz[t] = Z;
        tn[t] = in ? fma(Z, Z, fma(Y, Y, X * X)) : 1.0e300;        // (a padding particle never passes the test)
    }
    __syncthreads();
    int ii[AP2_R];
    double ax[AP2_R], ay[AP2_R], az[AP2_R], ti[AP2_R];
#pragma unroll
    for (int r = 0; r < AP2_R; r++) {
        ii[r] = bi * AP2_T + r * AP2_THREADS + (int)threadIdx.x;
        const bool in = ii[r] < n;
        const double X = in ? x[ii[r]] - ox : 0.0, Y = in ? y[ii[r]] - oy : 0.0, Z = in ? z[ii[r]] - oz : 0.0;
        ax[r] = -2.0 * X; ay[r] = -2.0 * Y; az[r] = -2.0 * Z;
        ti[r] = in ? thr - fma(Z, Z, fma(Y, Y, X * X)) : -1.0e300;
    }
    const bool diag = bi == bj;         // inside the diagonal tile only j < i counts
    for (int k0 = 0; k0 < AP2_T; k0 += AP2_CHUNK) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < AP2_CHUNK; u++) {
            const double jx = tx[k0 + u], jy = ty[k0 + u], jz = tz[k0 + u], jn = tn[k0 + u];
#pragma unroll
            for (int r = 0; r < AP2_R; r++) any |= fma(az[r], jz, fma(ay[r], jy, fma(ax[r], jx, jn))) < ti[r];
        }
        if (any) {
            for (int u = 0; u < AP2_CHUNK; u++) {
                const int j = j0 + k0 + u;
                const double jx = tx[k0 + u], jy = ty[k0 + u], jz = tz[k0 + u], jn = tn[k0 + u];
#pragma unroll
                for (int r = 0; r < AP2_R; r++)
                    if (fma(az[r], jz, fma(ay[r], jy, fma(ax[r], jx, jn))) < ti[r] && (diag ? j < ii[r] : true))
                        amc_push_candidate(ii[r], j, max_cand, cnt, D);
            }
        }
    }
}

hipError_t amc_launch_detect(amc_ctx *c)
{
    const long long n = c->n;
    if (n <= 0) return hipSuccess;
    const double cr2i = c->allpairs || c->detect_ap ? c->P.collision_range * c->P.collision_range * AMC_CR2_INFLATE : c->G.cr2_probe;
    amc_adj D;
    c->sweep_epoch = (c->sweep_epoch + 1u) & 0x3fffffffu;
    if (c->sweep_epoch == 0u) c->sweep_epoch = 1u;      // (0 is the value of the zero-initialised table)
    // launch plan of this sweep from the candidate count of the most recent sweep the host has seen (a word the resolve
    // kernel writes into host-mapped memory; it may lag by a step): a small sweep is committed by the ordered
    // workgroup itself, a large one by the wide commit kernel
    c->plan_split = !c->allpairs && !(c->h_host_ncand && *c->h_host_ncand <= c->plan_small);
    D.head = c->allpairs ? nullptr : c->W.adj_head; D.rec = c->W.cand4; D.sd = c->W.cand_s; D.epoch = c->sweep_epoch;
    D.sl_meta = c->W.sl_meta; D.sl_hits = c->W.sl_hits; D.ev_gen = c->W.ev_gen; D.mark = c->W.cand_mark;
    amc_prof_begin(c, AMC_K_DETECT);
    if (c->detect_ap) {
        if (n >= 16 * AP2_T && c->G.ncells > 0) {       // enough tiles to fill the chip (and a grid, whose origin and extent it uses): the register-tiled form
            const int ntiles = (int)((n + AP2_T - 1) / AP2_T);
            const long long nblocks = (long long)ntiles * (ntiles + 1) / 2;
            // coordinates relative to the grid origin; threshold raised by the cancellation bound of the expanded form
            const double ex = c->G.gx * c->G.h, ey = c->G.gy * c->G.h, ez = c->G.gz * c->G.h;
            const double thr = cr2i + 64.0 * 1.1102230246251565e-16 * (ex * ex + ey * ey + ez * ez);
            AMC_LAUNCH(c, k_detect_allpairs_tiled, dim3((unsigned)nblocks), dim3(AP2_THREADS), c->S.x, c->S.y,
                               c->S.z, (int)n, ntiles, thr, c->G.x0, c->G.y0, c->G.z0, c->W.max_cand, c->d_cnt, D);
        } else {
            const int ntiles = (int)((n + AP_T - 1) / AP_T);
            const long long nblocks = (long long)ntiles * (ntiles + 1) / 2;
            if (nblocks > 0)
                AMC_LAUNCH(c, k_detect_allpairs, dim3((unsigned)nblocks), dim3(AP_T), c->S.x, c->S.y,
                                   c->S.z, (int)n, ntiles, cr2i, c->W.max_cand, c->d_cnt, D);
        }
    } else {
        // (an overlapped run: four more blocks for the particles the fix-up kernel filed again under extra nodes)
        const bool extras = c->B.extra != nullptr && c->keep_K < 2;
        const int slot = (c->B.extra == c->extra_buf[1]) ? 1 : 0;
        static const int bs = getenv("AMC_DETECT_BS") ? atoi(getenv("AMC_DETECT_BS")) : 256;      // (experiments: 64 / 128 / 256)
        // (occupancy is not what bounds this kernel: capped at 4 waves per SIMD instead of 5 it takes the same 36.7 us at
        // N = 1e6, at 2 it takes 59 — it runs at the rate of its random requests; two or four particles per thread,
        // one after the other, change nothing at N = 1e6 and cost 5 / 15 us at N = 1e5)
        AMC_LAUNCH(c, k_detect_lists, dim3((unsigned)((n + bs - 1) / bs) + (extras ? 4u : 0u)), dim3(bs), c->G, c->B, n, cr2i,
                   c->G.cr_probe, c->W.max_cand, c->d_cnt, D, (const int *)(extras ? c->extra_count + slot : nullptr),
                   c->max_extra);
    }
    amc_prof_end(c);
    return hipGetLastError();
}

// multi-GPU: detection over [lo, hi) into this rank's candidate block, and the candidate graph from all blocks
hipError_t amc_launch_detect_own(amc_ctx *c)
{
    const long long cnt = c->hi - c->lo;
    amc_prof_begin(c, AMC_K_DETECT);
    if (cnt > 0)
        AMC_LAUNCH(c, k_detect_own, dim3((unsigned)((cnt + 255) / 256)), dim3(256), c->G, c->B, (long long)c->lo, (long long)c->hi,
                   c->G.cr2_probe, c->G.cr_probe, c->cand_send, c->cand_cap, c->d_cnt);
    amc_prof_end(c);
    return hipGetLastError();
}

hipError_t amc_launch_ingest(amc_ctx *c, int world)
{
    amc_adj D;
    c->sweep_epoch = (c->sweep_epoch + 1u) & 0x3fffffffu;
    if (c->sweep_epoch == 0u) c->sweep_epoch = 1u;
    c->plan_split = !(c->h_host_ncand && *c->h_host_ncand <= c->plan_small);
    D.head = c->W.adj_head; D.rec = c->W.cand4; D.sd = c->W.cand_s; D.epoch = c->sweep_epoch;
    D.sl_meta = c->W.sl_meta; D.sl_hits = c->W.sl_hits; D.ev_gen = c->W.ev_gen; D.mark = c->W.cand_mark;
    amc_prof_begin(c, AMC_K_DETECT);
    AMC_LAUNCH(c, k_ingest_candidates, dim3(16), dim3(256), (const int *)c->cand_recv, world, 2 + 2 * c->cand_cap, c->cand_cap,
               c->W.max_cand, c->d_cnt, D, c->cand_send);
    amc_prof_end(c);
    return hipGetLastError();
}
