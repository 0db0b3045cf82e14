// amc_resolve.hip — the ordered half of the p-p sweep: turns the candidate pairs into exactly the collisions the
// reference's sequential, in-place cell loops would perform (Pore:160-255 driven by Pore:520-549, Cube:231-336).
//
// Why this is not a plain parallel-for: the reference updates positions/velocities in place while it scans, cells
// overlap (a pair can be visited in up to 8 colour groups / cells) and a particle hit twice in one sweep sees the
// first collision's result.  Collisions are rare (~N/2000 per step) and almost always isolated,
// so the work is organised as OPTIMISTIC CLUSTERS WITH CONSERVATIVE VALIDATION:
//
//   1. clusters = connected components of the candidate graph (label propagation, label = lowest particle index);
//   2. every cluster gets a LITERAL emulation of the reference restricted to its members — same colour-group / cell
//      order, same membership predicates evaluated at the same moments (Pore:527-530; Cube's in_x/in_y/in_z masks
//      are evaluated per x-layer / (x,y)-layer / cell, Cube:233-238), same i>j loop order (members in ascending
//      particle index), same arithmetic (amc_collide) — on a scratch copy of the state.
//        * two-particle clusters (~99 %): one thread, both particles in registers, no sorting;
//        * larger clusters: members sorted by (label, index) with a small bitonic sort, working set in LDS;
//   3. validation: every position a member occupied during the emulation is checked against all particles outside
//      its cluster (pre-sweep positions via the detection grid, other clusters' new positions via an overlay list
//      per grid cell).  Anything within collision_range*(1+1e-9) merges the clusters / pulls the particle in, and
//      the round is repeated from the untouched pre-sweep state.  When a round validates, no test the reference
//      performs between a member and a non-member can hit, hence the restricted emulation is exactly what the
//      reference computes;
//   4. commit: scratch state -> particle arrays, completed paths -> histogram / record buffer, counters.
//
// The ordered logic runs in ONE 512-thread workgroup (phases separated by __syncthreads, counters in LDS): the data
// set is a few hundred pairs, so the cost is latency, not throughput.  Everything that is scattered memory traffic
// is kept off that single CU: detect gathers the candidates' state into a SoA table (coalesced reads here), and the
// first round's validation probes and the commit scatter are separate wide kernels.
#include "amc_grid_dev.h"

#define RS_T 512
#define RS_SORT_LDS 2048       // complex-cluster members sorted in LDS up to this many
#define RS_POOL 256            // complex-cluster working set held in LDS up to this many members
#define RS_MAX_ROUNDS 256
#define AMC_CR2_INFLATE (1.0 + 1.0e-9)

struct rs_args {
    amc_params P;
    amc_state S;
    amc_grid G;
    amc_lists B;
    amc_resolve_ws W;
    amc_out O;
    long long n;
    int allpairs;
    int single_round;         // multi-GPU: a continuation launch runs exactly one round and hands back to the host
    int defer_commit;         // leave the results in the slot arrays: the next streaming pass (or k_apply) writes them
    int apply_only;           // k_commit: only write deferred results (amc_flush)
    int allow_mono;           // small sweeps may run validation + commit inside resolve_A
    int force_mono;           // the host launched ONLY this kernel (it expects a small sweep): do everything here
    int *host_ncand;          // host-mapped word: candidate count of this sweep, read (lagging) by the host to pick the launch plan (saves three kernels' latency)
    int count_pp;             // this rank adds the sweep's collision count to the counters (rank 0 in multi-GPU)
    long long lo, hi;         // owned particle range: completed paths are emitted by the owner of the particle
    double inv_dx, inv_dy, inv_dz;   // 1/dx.. for floor() GUESSES only (membership is decided by the exact comparisons)
    long long *dbg;           // optional [16] phase timers (wall_clock64 ticks, 100 MHz), diagnostic only
};

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
    W.sl_x[s] = q.x; W.sl_y[s] = q.y; W.sl_z[s] = q.z; W.sl_vx[s] = q.vx; W.sl_vy[s] = q.vy; W.sl_vz[s] = q.vz;
    W.sl_d[s] = q.d; W.sl_dx[s] = q.dx; W.sl_dy[s] = q.dy; W.sl_dz[s] = q.dz; W.sl_flag[s] = q.flag ? 1 : 0;
    W.sl_moved[s] = 1;
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

// one hit inside an emulation: resolve p1 (= j, lower index) / p2 (= i) in registers, log events + history.
// Returns true if the particles moved.
AMC_DEV bool rs_hit(const rs_args &A, rs_shared *sh, amc_particle &p1, amc_particle &p2, int pj, int pi, int sj,
                    int si, int phase, long long cell)
{
    const amc_resolve_ws &W = A.W;
    auto emit = [&](int which, double tot, double px, double py, double pz) {
        const int e = atomicAdd(&sh->nev, 1);
        if (e < W.max_events) {
            W.ev_phase[e] = phase; W.ev_cell[e] = cell; W.ev_i[e] = pi; W.ev_j[e] = pj; W.ev_which[e] = which;
            W.ev_gen[e] = sh->cur_round; W.ev_slot[e] = si;
            W.ev_val[4 * e + 0] = tot; W.ev_val[4 * e + 1] = px; W.ev_val[4 * e + 2] = py; W.ev_val[4 * e + 3] = pz;
        } else {
            sh->ovf = 1;
        }
    };
    long long tq__ = (A.dbg && threadIdx.x == 0) ? wall_clock64() : 0;
    const int fail__ = amc_collide(p1, p2, A.P.collision_range, A.P.argon_mass, emit);
    if (A.dbg && threadIdx.x == 0) { if (p1.x == 1.2345e300) A.dbg[15] = 3; A.dbg[12] += wall_clock64() - tq__; }
    if (fail__) {
        atomicAdd(&sh->nfp, 1);     // the reference would raise FloatingPointError here (Pore:11,185)
        return false;
    }
    atomicAdd(&W.sl_hits[si], 1);   // (no value needed back: the thread does not wait for the memory round trip)
    const int h = atomicAdd(&sh->nhist, 2);
    if (h + 1 < W.max_hist) {
        W.hist[h] = rs_hist_make(p1.x, p1.y, p1.z, sj, sh->cur_round);
        W.hist[h + 1] = rs_hist_make(p2.x, p2.y, p2.z, si, sh->cur_round);
    } else {
        sh->ovf = 1;
    }
    return true;
}

AMC_DEV int rs_pore_cell(const amc_params &P, double x, double y, double z, int gx, int gy, int gz)
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
AMC_DEV void rs_axis_k(double v, double d, double inv_d, double ov, int &ka, int &kb)
{
    ka = kb = (int)0x80000000;
    const double f = floor(v * inv_d);
    if (!(f > -1.0e9 && f < 1.0e9)) return;
    for (int dk = -1; dk <= 1; dk++) {
        const int k = (int)f + dk;
        const double lo = (double)k * d - ov, hi = (double)(k + 1) * d;
        if (lo < v && v < hi) { if (ka == (int)0x80000000) ka = k; else kb = k; }
    }
}
AMC_DEV void rs_pore_ks(const rs_args &A, const amc_particle &q, int *k)
{
    const amc_params &P = A.P;
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
AMC_DEV int rs_pore_cell_k(const amc_params &P, const int *k, int g)
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
AMC_DEV amc_particle rs_load_cst(const amc_resolve_ws &W, int k, int which)
{
    const size_t m = (size_t)W.max_cand;
    const double *t = W.cst + (size_t)(11 * which) * m + k;
    amc_particle q;
    q.x = t[0 * m]; q.y = t[1 * m]; q.z = t[2 * m]; q.vx = t[3 * m]; q.vy = t[4 * m]; q.vz = t[5 * m];
    q.d = t[6 * m]; q.dx = t[7 * m]; q.dy = t[8 * m]; q.dz = t[9 * m]; q.flag = t[10 * m] != 0.0;
    return q;
}

template <int GEOM>
AMC_DEV void rs_emulate_pair(const rs_args &A, rs_shared *sh, int k, int pj, int pi, int sj, int si)
{
    const amc_params &P = A.P;
    long long t0__ = (A.dbg && threadIdx.x == 0) ? wall_clock64() : 0;
    amc_particle p1 = rs_load_cst(A.W, k, 0), p2 = rs_load_cst(A.W, k, 1);   // coalesced rows gathered by detect
    if (A.dbg && threadIdx.x == 0) { if (p1.x + p2.x == 1.2345e300) A.dbg[15] = 1; const long long t1__ = wall_clock64(); A.dbg[13] += t1__ - t0__; t0__ = t1__; }
    bool moved = false;
    const double cr = P.collision_range;
    if (GEOM == AMC_GEOM_CELL) {
        if (amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr)) moved |= rs_hit(A, sh, p1, p2, pj, pi, sj, si, 16, 0);
    } else if (GEOM == AMC_GEOM_CUBE) {
        // Cube:231-238.  A coordinate lies in at most two overlapping layers, so the layers holding BOTH members are
        // enumerated directly (rs_next_common) instead of walking all nx*ny*nz cells — lanes of a wave would each
        // enter the nested loops at different iterations and the wave would execute the whole nest.
        // The stale masks come out of the structure: the x test is made once when an x-layer starts, the y test
        // once per (x,y)-layer, the z test per cell, each from the state at that moment (in_x_layer / in_y_layer /
        // in_z_layer); none is re-evaluated after a hit inside the layer.
        for (int lx = rs_next_common(p1.x, p2.x, P.dx, A.inv_dx, P.overlap_x, P.nx, 0); lx >= 0;
             lx = rs_next_common(p1.x, p2.x, P.dx, A.inv_dx, P.overlap_x, P.nx, lx + 1))
            for (int ly = rs_next_common(p1.y, p2.y, P.dy, A.inv_dy, P.overlap_y, P.ny, 0); ly >= 0;
                 ly = rs_next_common(p1.y, p2.y, P.dy, A.inv_dy, P.overlap_y, P.ny, ly + 1))
                for (int lz = rs_next_common(p1.z, p2.z, P.dz, A.inv_dz, P.overlap_z, P.nz, 0); lz >= 0;
                     lz = rs_next_common(p1.z, p2.z, P.dz, A.inv_dz, P.overlap_z, P.nz, lz + 1))
                    if (amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr))
                        moved |= rs_hit(A, sh, p1, p2, pj, pi, sj, si, 16, ((long long)lx * P.ny + ly) * P.nz + lz);
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
            if (rs_hit(A, sh, p1, p2, pj, pi, sj, si, 16 + hit_g, hit_c)) {
                moved = true;
                rs_pore_ks(A, p1, k1);
                rs_pore_ks(A, p2, k2);
                ov = amc_overlap(p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, cr);
            }
            g = hit_g + 1;
        }
    }
    if (moved) {
        rs_store_slot(A.W, sj, p1);
        rs_store_slot(A.W, si, p2);
    }
}

// ---- generic cluster (3+ members): literal emulation on the working set [b,e) ----------------------
AMC_DEV void rs_test_work(const rs_args &A, rs_shared *sh, const rs_work &K, int wj, int wi, int phase, long long cell)
{
    if (!amc_overlap(K.x[wj], K.y[wj], K.z[wj], K.x[wi], K.y[wi], K.z[wi], A.P.collision_range)) return;
    amc_particle p1 = rs_load_work(K, wj), p2 = rs_load_work(K, wi);
    const int pj = K.pidx[wj], pi = K.pidx[wi];
    if (rs_hit(A, sh, p1, p2, pj, pi, K.slot[wj], K.slot[wi], phase, cell)) {
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

AMC_DEV void rs_emulate_generic(const rs_args &A, rs_shared *sh, const rs_work &K, int b, int e)
{
    const amc_params &P = A.P;
    if (P.geometry == AMC_GEOM_CELL) {
        for (int a = b + 1; a < e; a++)                                                     // Pore:168-169
            for (int c = b; c < a; c++) rs_test_work(A, sh, K, c, a, 16, 0);
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
                            if (K.tmp[c] == 7) rs_test_work(A, sh, K, c, a, 16, cell);
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
                    if (K.tmp[c] == ca) rs_test_work(A, sh, K, c, a, 16 + g, ca);
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
    int *p, *label, *size;
    int cap;
};

// get (or create) the slot of particle p; creation is published by the barrier / kernel boundary that follows
AMC_DEV void rs_claim_slot(const amc_resolve_ws &W, rs_shared *sh, int cap, int p)
{
    const int old = atomicCAS(&W.slot_of[p], -1, -2);
    if (old == -1) {
        const int s = atomicAdd(&sh->nslots, 1);
        if (s < cap) {
            W.sl_p[s] = p;
            W.slot_of[p] = s;
        } else {
            sh->ovf = 1;
            W.slot_of[p] = -1;
        }
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
AMC_DEV void rs_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
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

AMC_DEV void rs_emulate_coop(const rs_args &A, rs_shared *sh, const rs_work &K, int b, int e)
{
    const amc_params &P = A.P;
    const int lane = threadIdx.x & 63, m = e - b;
    const int w = b + lane;                 // my member (lanes < m)
    int pa, pc;
    rs_lane_pair(lane, m, pa, pc);
    if (P.geometry == AMC_GEOM_CELL) {
        if (lane == 0)
            for (int a = b + 1; a < e; a++)                                                 // Pore:168-169
                for (int c = b; c < a; c++) rs_test_work(A, sh, K, c, a, 16, 0);
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
                                if (K.tmp[c] == 7) rs_test_work(A, sh, K, c, a, 16, cell);
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
                        if (K.tmp[c] == ca) rs_test_work(A, sh, K, c, a, 16 + g, ca);
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
    const int sme = rs_hist_slot(me);
    const int pme = W.sl_p[sme];
    const int lme = label[sme];
    const double x = me.x, y = me.y, z = me.z;
    // only the cells overlapped by the collision_range box around the new position can hold a partner (2 to 3 on
    // average): fetch their list heads and overlay heads first, then the entries
    int c_lo[4], c_hi[4], lh[8], ovh[8];
    const int ncell = amc_grid_box_ranges(G, x, y, z, A.P.collision_range * 1.000001, c_lo, c_hi);
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
    double4 r0[RS_PF], o0[RS_PF];
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
    auto grid_entry = [&](int idx, const double4 &r) {
        if (idx == pme) return;
        const double ax = r.x - x, ay = r.y - y, az = r.z - z;
        if (ax * ax + ay * ay + az * az < cr2i) {
            const int so = W.slot_of[idx];
            if (so >= 0 && so < ns && label[so] == lme) return;
            if (so < 0) rs_claim_slot(W, cnt, cap, idx);
            rs_add_edge(W, cnt, pme, idx);
        }
    };
    auto overlay_entry = [&](const double4 &o) {
        const int s2 = rs_hist_slot(o);
        if (rs_hist_gen(o) != W.sl_gen[s2]) return;       // position of an emulation that was redone since
        if (label[s2] == lme) return;
        const double ax = o.x - x, ay = o.y - y, az = o.z - z;
        if (ax * ax + ay * ay + az * az < cr2i) rs_add_edge(W, cnt, pme, W.sl_p[s2]);
    };
    // pre-sweep positions of the particles binned into those cells
#pragma unroll
    for (int k = 0; k < RS_PF; k++)
        if (lh[k] >= 0) {
            grid_entry(lh[k], r0[k]);
            for (int q = amc_rec_next(r0[k]); q >= 0;) {
                const double4 r = A.B.rec[q];
                grid_entry(q, r);
                q = amc_rec_next(r);
            }
        }
    for (int k = RS_PF; k < 2 * ncell; k++)
        for (int q = lh[k]; q >= 0;) {
            const double4 r = A.B.rec[q];
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

#define RS_STAMP(slot)                                                                     \
    do {                                                                                   \
        if (A.dbg && tid == 0) {                                                           \
            const long long now__ = wall_clock64();                                        \
            A.dbg[slot] += now__ - t_last;                                                 \
            t_last = now__;                                                                \
        }                                                                                  \
    } while (0)

#define RS_NS 6144             // slot labels / sizes / dirty flags kept in LDS (54 KB)
#define RS_LAY 4096            // ints of the grid's layer tables kept in LDS
#define RS_SMALL 640           // candidate pairs up to which resolve_A does everything itself

// MODE 0: first round only (claim, label, emulate), validation + commit are the wide kernels that follow
// MODE 1: continuation: if the wide validation found merges, run the remaining rounds (validation in-kernel)
// MODE 2: everything in one kernel incl. brute-force validation and commit (no detection grid: single cells, small N)
template <int GEOM, int MODE>
__global__ __launch_bounds__(RS_T) void k_resolve(rs_args A)
{
    long long t_last = (A.dbg && threadIdx.x == 0) ? wall_clock64() : 0;
    __shared__ rs_shared sh;
    __shared__ unsigned long long lds_keys[RS_SORT_LDS];
    __shared__ double pool_d[10][RS_POOL];
    __shared__ int pool_tmp[RS_POOL], pool_pidx[RS_POOL], pool_slot[RS_POOL];
    __shared__ uint8_t pool_flag[RS_POOL], pool_moved[RS_POOL];
    __shared__ int s_label[RS_NS], s_size[RS_NS];
    __shared__ unsigned char s_dirty[RS_NS];
    __shared__ int s_lay[RS_LAY];
    const amc_resolve_ws &W = A.W;
    const int tid = threadIdx.x;
    amc_dev_counters *cnt = A.O.cnt;
    rs_shared *ctl = (rs_shared *)W.ctl;
    int ncand;
    if (MODE == 1) {
        if (tid == 0) sh = *ctl;
        __syncthreads();
        if (!sh.active || !sh.dirty || sh.ovf) return;      // the first round validated (or nothing to do)
        ncand = sh.ncand;
    } else {
        ncand = (int)cnt->cand_count;
        if (ncand > W.max_cand) ncand = W.max_cand;
        __syncthreads();
        if (tid == 0) {
            sh.nslots = 0; sh.nedges = 0; sh.nhist = 0; sh.nev = 0; sh.dirty = 0; sh.changed = 0; sh.nhits = 0;
            sh.nfp = 0; sh.ovf = 0; sh.nclusters = 0; sh.ncomplex = 0; sh.rounds = 0; sh.ncand = ncand;
            sh.active = ncand > 0; sh.ok = 1; sh.edges_done = 0; sh.hist_begin = 0; sh.cur_round = 0; sh.nslots0 = 0;
            sh.lazy_ns = 0;         // the streaming pass before this sweep consumed the previous sweep's deferred results
            cnt->cand_count = 0;
            if (A.host_ncand) *A.host_ncand = ncand;
            if (ncand == 0) *ctl = sh;
        }
        __syncthreads();
        if (ncand == 0) return;     // uniform: nothing to resolve this sweep
    }

    // MODE 2 always, MODE 0 for small sweeps: validation and commit in this kernel (the wide kernels then find
    // ctl.active == 0 and exit); large sweeps hand over after the first round
    const bool mono = (MODE == 2) || (MODE == 0 && A.allow_mono && (A.force_mono || ncand <= RS_SMALL));
    const double cr2i = A.P.collision_range * A.P.collision_range * AMC_CR2_INFLATE;
    rs_slots V;
    V.p = W.sl_p;
    unsigned char *vdirty;
    // (the slot arrays in global memory hold W.max_slots entries: labels in LDS must not let validation claim more)
    if ((MODE == 1 ? sh.nslots : 2 * ncand) + 256 <= RS_NS) { V.label = s_label; V.size = s_size; V.cap = RS_NS < W.max_slots ? RS_NS : W.max_slots; vdirty = s_dirty; }
    else { V.label = W.sl_label; V.size = W.sl_tmp; V.cap = W.max_slots; vdirty = W.sl_dirty; }
    // grid layer tables -> LDS (every validation probe reads them)
    amc_grid G = A.G;
    if ((MODE != 0 || mono) && !A.allpairs && 3 * G.gz <= RS_LAY) {
        for (int k = tid; k < 3 * G.gz; k += RS_T) s_lay[k] = A.G.lay_lo[k];     // the three tables are contiguous
        G.lay_lo = s_lay; G.lay_n = s_lay + G.gz; G.lay_off = s_lay + 2 * G.gz;
    }
    __syncthreads();

    if (MODE != 1) {
        // ---- slots for the candidate endpoints; candidates become slot pairs -------------------------------------------
        for (int k = tid; k < ncand; k += RS_T) {
            rs_claim_slot(W, &sh, V.cap, W.cand_i[k]);
            rs_claim_slot(W, &sh, V.cap, W.cand_j[k]);
        }
        __syncthreads();
        for (int k = tid; k < ncand; k += RS_T) {
            W.cand_si[k] = W.slot_of[W.cand_i[k]];
            W.cand_sj[k] = W.slot_of[W.cand_j[k]];
        }
        __syncthreads();
    }
    RS_STAMP(0);

    int rounds = sh.rounds;
    int edges_done = sh.edges_done;     // edges [0, edges_done) already hold slot ids
    for (;;) {
        rounds++;
        const int ns = sh.nslots < V.cap ? sh.nslots : V.cap;
        const int nedges = sh.nedges < W.max_edges ? sh.nedges : W.max_edges;
        __syncthreads();
        // ---- per-round reset; new merge edges: particle ids -> slot ids -------------------------------------------------
        for (int s = tid; s < ns; s += RS_T) {
            V.label[s] = s;
            V.size[s] = 0;
            vdirty[s] = (rounds == 1);          // round 1 emulates everything; later rounds only what the new edges touch
        }
        for (int k = edges_done + tid; k < nedges; k += RS_T) {
            W.edge_a[k] = W.slot_of[W.edge_a[k]];
            W.edge_b[k] = W.slot_of[W.edge_b[k]];
        }
        const int edges_new = edges_done;
        edges_done = nedges;
        if (tid == 0) { sh.dirty = 0; sh.nclusters = 0; sh.ncomplex = 0; sh.cur_round = rounds; sh.hist_begin = sh.nhist < W.max_hist ? sh.nhist : W.max_hist; }
        __syncthreads();
        // ---- connected components by label propagation (label = lowest slot id of the cluster) -------------------------
        for (;;) {
            int changed = 0;
            for (int k = tid; k < ncand + nedges; k += RS_T) {
                const int sa = k < ncand ? W.cand_si[k] : W.edge_a[k - ncand];
                const int sb = k < ncand ? W.cand_sj[k] : W.edge_b[k - ncand];
                if (sa < 0 || sb < 0 || sa >= ns || sb >= ns) continue;
                const int la = V.label[sa], lb = V.label[sb];
                if (la < lb) { atomicMin(&V.label[sb], la); changed = 1; }
                else if (lb < la) { atomicMin(&V.label[sa], lb); changed = 1; }
            }
            if (!__syncthreads_or(changed)) break;
        }
        // ---- cluster sizes, accumulated on the label slot ------------------------------------------------------------------
        for (int s = tid; s < ns; s += RS_T) atomicAdd(&V.size[V.label[s]], 1);
        // clusters touched by the merge edges of the previous validation are re-emulated; everything else keeps its
        // results, events and history (tagged with the round they were produced in)
        for (int k = edges_new + tid; k < nedges; k += RS_T) {
            const int sa = W.edge_a[k], sb = W.edge_b[k];
            if (sa >= 0 && sa < ns) vdirty[V.label[sa]] = 1;
            if (sb >= 0 && sb < ns) vdirty[V.label[sb]] = 1;
        }
        __syncthreads();
        for (int s = tid; s < ns; s += RS_T)
            if (vdirty[V.label[s]]) { W.sl_moved[s] = 0; W.sl_gen[s] = rounds; W.sl_hits[s] = 0; }
        __syncthreads();
        RS_STAMP(1);
        // ---- members of clusters with 3+ particles are collected for the generic path ----------------------------------------
        unsigned long long *keys = lds_keys;
        for (int s = tid; s < ns; s += RS_T) {
            if (V.label[s] == s) atomicAdd(&sh.nclusters, 1);
            if (V.size[V.label[s]] >= 3 && vdirty[V.label[s]]) {
                const int k = atomicAdd(&sh.ncomplex, 1);
                const unsigned long long key = ((unsigned long long)(unsigned)V.label[s] << 32) | (unsigned)V.p[s];
                if (k < RS_SORT_LDS) keys[k] = key; else W.sl_key[k] = key;
            }
        }
        __syncthreads();
        RS_STAMP(6);
        const int nc = sh.ncomplex;
        // Usually there are only a few such members: then ONE wave runs their whole pipeline (rank sort, load, literal
        // emulation, write-back) with wave-level synchronisation while the other seven waves do the two-particle
        // clusters — the two kinds of cluster are disjoint, so the phases overlap instead of following each other.
        const bool split = nc > 0 && nc <= 64 && nc <= RS_POOL;
        rs_work K;
        K.x = pool_d[0]; K.y = pool_d[1]; K.z = pool_d[2]; K.vx = pool_d[3]; K.vy = pool_d[4]; K.vz = pool_d[5];
        K.d = pool_d[6]; K.dx = pool_d[7]; K.dy = pool_d[8]; K.dz = pool_d[9];
        K.tmp = pool_tmp; K.pidx = pool_pidx; K.slot = pool_slot; K.flag = pool_flag; K.moved = pool_moved;
        if (split && tid < 64) {
            unsigned long long *sorted = lds_keys + RS_SORT_LDS / 2;
            const int w = tid;
            unsigned long long mykey = 0;
            int rank = 0;
            if (w < nc) {
                mykey = keys[w];
                for (int k = 0; k < nc; k++) rank += keys[k] < mykey;          // keys are unique: rank sort
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (w < nc) sorted[rank] = mykey;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (w < nc) {
                const int p = (int)(sorted[w] & 0xffffffffULL);
                const amc_particle q = rs_load_particle(A.S, p);
                rs_store_work(K, w, q);
                K.moved[w] = 0;
                K.pidx[w] = p;
                K.slot[w] = W.slot_of[p];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (A.dbg && tid == 0) { const long long n__ = wall_clock64(); A.dbg[2] += n__ - t_last; t_last = n__; }
            // One or two clusters (the usual case): one after the other, each by the whole wave (rs_emulate_coop; uniform
            // control flow, every lane walks the same list).  More: one lane per cluster head — the lanes then run the
            // same code side by side on different clusters, which beats taking the clusters in turn.
            const bool is_head = w < nc && !(w > 0 && (unsigned)(sorted[w - 1] >> 32) == (unsigned)(sorted[w] >> 32));
            const int nheads = __popcll(__ballot(is_head));
            if (nheads <= 2) {
                for (int w0 = 0; w0 < nc;) {
                    const unsigned lab = (unsigned)(sorted[w0] >> 32);
                    int e = w0 + 1;
                    while (e < nc && (unsigned)(sorted[e] >> 32) == lab) e++;
                    if (e - w0 >= 2) {
                        if (e - w0 <= RS_COOP_MAX) rs_emulate_coop(A, &sh, K, w0, e);
                        else if (w == 0) rs_emulate_generic(A, &sh, K, w0, e);
                    }
                    w0 = e;
                }
            } else if (is_head) {
                const unsigned lab = (unsigned)(sorted[w] >> 32);
                int e = w + 1;
                while (e < nc && (unsigned)(sorted[e] >> 32) == lab) e++;
                if (e - w >= 2) rs_emulate_generic(A, &sh, K, w, e);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (A.dbg && tid == 0) { const long long n__ = wall_clock64(); A.dbg[3] += n__ - t_last; t_last = n__; }
            if (w < nc && K.moved[w]) rs_store_slot(W, K.slot[w], rs_load_work(K, w));
        } else {
            // ---- two-particle clusters straight from the candidate list, both particles in registers ------------------------
            const int t0 = split ? tid - 64 : tid, tstride = split ? RS_T - 64 : RS_T;
            for (int k = t0; k < ncand; k += tstride) {
                const int si = W.cand_si[k], sj = W.cand_sj[k];
                if (si < 0 || sj < 0 || si >= ns || sj >= ns) continue;
                if (V.size[V.label[si]] != 2 || !vdirty[V.label[si]]) continue;
                rs_emulate_pair<GEOM>(A, &sh, k, W.cand_j[k], W.cand_i[k], sj, si);
            }
        }
        RS_STAMP(7);
        __syncthreads();
        RS_STAMP(2);
        // ---- larger clusters: sort members by (label, index), working set in LDS when it fits ----------------------------
        if (nc > 0 && !split) {
            int m = 1;
            while (m < nc) m <<= 1;
            if (nc > RS_SORT_LDS) {
                for (int k = tid; k < RS_SORT_LDS; k += RS_T) W.sl_key[k] = lds_keys[k];
                keys = W.sl_key;
            }
            for (int k = nc + tid; k < m; k += RS_T) keys[k] = ~0ULL;
            __syncthreads();
            rs_bitonic(keys, m);
            if (nc > RS_POOL) {
                K.x = W.cw_d[0]; K.y = W.cw_d[1]; K.z = W.cw_d[2]; K.vx = W.cw_d[3]; K.vy = W.cw_d[4]; K.vz = W.cw_d[5];
                K.d = W.cw_d[6]; K.dx = W.cw_d[7]; K.dy = W.cw_d[8]; K.dz = W.cw_d[9];
                K.tmp = W.cw_tmp; K.pidx = W.cw_pidx; K.slot = W.cw_slot; K.flag = W.cw_flag; K.moved = W.cw_moved;
            }
            for (int w = tid; w < nc; w += RS_T) {
                const int p = (int)(keys[w] & 0xffffffffULL);
                const amc_particle q = rs_load_particle(A.S, p);
                rs_store_work(K, w, q);
                K.moved[w] = 0;
                K.pidx[w] = p;
                K.slot[w] = W.slot_of[p];
            }
            __syncthreads();
            for (int w = tid; w < nc; w += RS_T) {
                const unsigned lab = (unsigned)(keys[w] >> 32);
                if (w > 0 && (unsigned)(keys[w - 1] >> 32) == lab) continue;      // not a cluster head
                int e = w + 1;
                while (e < nc && (unsigned)(keys[e] >> 32) == lab) e++;
                if (e - w >= 2) rs_emulate_generic(A, &sh, K, w, e);
            }
            __syncthreads();
            for (int w = tid; w < nc; w += RS_T)
                if (K.moved[w]) rs_store_slot(W, K.slot[w], rs_load_work(K, w));
        }
        __syncthreads();
        RS_STAMP(3);
        const int nh = sh.nhist < W.max_hist ? sh.nhist : W.max_hist;
        if (!mono && (MODE == 0 || (MODE == 1 && A.single_round))) {
            // ---- hand over to the wide validation kernel: labels to global memory, history into the overlay -------------
            if (V.label != W.sl_label)
                for (int s = tid; s < ns; s += RS_T) W.sl_label[s] = V.label[s];
            for (int h = sh.hist_begin + tid; h < nh; h += RS_T) W.ov_next[h] = atomicExch(&W.ov_head[rs_hist_cell(A, G, h)], h);
            __syncthreads();
            if (tid == 0) { sh.rounds = rounds; sh.edges_done = edges_done; sh.nslots0 = ns; *ctl = sh; }
            RS_STAMP(4);
            if (A.dbg && tid == 0) { A.dbg[8] += 1; A.dbg[9] += ncand; A.dbg[10] += sh.ncomplex; A.dbg[11] += 1; }
            return;
        }
        // ---- validate: every new position against everything outside its cluster ------------------------------------------------
        if (!A.allpairs) {
            for (int h = sh.hist_begin + tid; h < nh; h += RS_T) W.ov_next[h] = atomicExch(&W.ov_head[rs_hist_cell(A, G, h)], h);
            __syncthreads();
            if (A.dbg && tid == 0) { const long long n__ = wall_clock64(); A.dbg[14] += n__ - t_last; }
            for (int h = sh.hist_begin + tid; h < nh; h += RS_T) rs_probe(A, G, &sh, V.label, ns, V.cap, h, cr2i);
        } else {
            // no grid (single cell / small N): brute force against all particles and all history entries
            const int hb = sh.hist_begin;
            for (long long w = tid; w < (long long)(nh - hb) * A.n; w += RS_T) {
                const int h = hb + (int)(w / A.n);
                const int idx = (int)(w % A.n);
                const double4 hr = W.hist[h];
                const int sme = rs_hist_slot(hr);
                if (idx == V.p[sme]) continue;
                const double ex = A.S.x[idx] - hr.x, ey = A.S.y[idx] - hr.y, ez = A.S.z[idx] - hr.z;
                if (ex * ex + ey * ey + ez * ez < cr2i) {
                    const int so = W.slot_of[idx];
                    if (so >= 0 && so < ns && V.label[so] == V.label[sme]) continue;
                    if (so < 0) rs_claim_slot(W, &sh, V.cap, idx);
                    rs_add_edge(W, &sh, V.p[sme], idx);
                }
            }
            for (long long w = tid; w < (long long)(nh - hb) * nh; w += RS_T) {
                const int h = hb + (int)(w / nh), h2 = (int)(w % nh);
                if (h2 >= h) continue;
                const double4 ha = W.hist[h], hb2 = W.hist[h2];
                const int s1 = rs_hist_slot(ha), s2 = rs_hist_slot(hb2);
                if (rs_hist_gen(hb2) != W.sl_gen[s2]) continue;
                if (V.label[s1] == V.label[s2]) continue;
                const double ex = hb2.x - ha.x, ey = hb2.y - ha.y, ez = hb2.z - ha.z;
                if (ex * ex + ey * ey + ez * ez < cr2i) rs_add_edge(W, &sh, V.p[s1], V.p[s2]);
            }
        }
        __syncthreads();
        RS_STAMP(4);
        if (!sh.dirty || sh.ovf || rounds >= RS_MAX_ROUNDS) break;
        __syncthreads();
    }

    const bool ok = !sh.ovf && !(sh.dirty && rounds >= RS_MAX_ROUNDS);
    if (MODE == 1) {
        // the wide commit kernel finishes the sweep
        __syncthreads();
        if (tid == 0) { sh.rounds = rounds; sh.edges_done = edges_done; sh.ok = ok; sh.dirty = 0; *ctl = sh; }
        if (A.dbg && tid == 0) { A.dbg[8] += rounds - 1; A.dbg[12] += 1; }
        return;
    }
    // ---- commit (monolithic path) ---------------------------------------------------------------------------------------
    const int ns = sh.nslots < V.cap ? sh.nslots : V.cap;
    if (!A.allpairs) {
        const int nh_all = sh.nhist < W.max_hist ? sh.nhist : W.max_hist;
        for (int h = tid; h < nh_all; h += RS_T) W.ov_head[rs_hist_cell(A, G, h)] = -1;
    }
    const bool defer = ok && A.defer_commit;        // the next streaming pass reads the slot arrays through slot_of[]
    for (int s = tid; s < ns && !defer; s += RS_T) {
        const int p = V.p[s];
        if (ok && W.sl_moved[s]) {
            A.S.x[p] = W.sl_x[s]; A.S.y[p] = W.sl_y[s]; A.S.z[p] = W.sl_z[s];
            A.S.vx[p] = W.sl_vx[s]; A.S.vy[p] = W.sl_vy[s]; A.S.vz[p] = W.sl_vz[s];
            A.S.d[p] = W.sl_d[s]; A.S.dx[p] = W.sl_dx[s]; A.S.dy[p] = W.sl_dy[s]; A.S.dz[p] = W.sl_dz[s];
            A.S.flag[p] = W.sl_flag[s];
        }
        W.slot_of[p] = -1;
    }
    if (tid == 0) sh.nhits = 0;
    __syncthreads();
    if (ok) {
        const int nev = sh.nev < W.max_events ? sh.nev : W.max_events;
        for (int e = tid; e < nev; e += RS_T) {
            if (W.ev_gen[e] != W.sl_gen[W.ev_slot[e]]) continue;            // event of an emulation that was redone
            const int owner = W.ev_which[e] ? W.ev_i[e] : W.ev_j[e];
            if (owner < A.lo || owner >= A.hi) continue;
            amc_emit(A.O, W.ev_phase[e], W.ev_cell[e], W.ev_i[e], W.ev_j[e], W.ev_which[e], W.ev_val[4 * e + 0],
                     W.ev_val[4 * e + 1], W.ev_val[4 * e + 2], W.ev_val[4 * e + 3]);
        }
        for (int s2 = tid; s2 < ns; s2 += RS_T) {
            // the counts were updated by atomics (performed in L2): read them there too, not from this CU's L1
            const int hs = __hip_atomic_load(&W.sl_hits[s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (hs) atomicAdd(&sh.nhits, hs);
        }
    }
    __syncthreads();
    RS_STAMP(5);
    if (tid == 0) {
        if (A.dbg) { A.dbg[8] += rounds; A.dbg[9] += ncand; A.dbg[10] += sh.ncomplex; A.dbg[11] += 1; }
        cnt->n_candidates += (unsigned long long)ncand;
        cnt->n_clusters += (unsigned long long)sh.nclusters;
        cnt->n_rounds += (unsigned long long)rounds;
        if (ok) {
            cnt->n_pp += (unsigned long long)sh.nhits;
            cnt->n_fp_errors += (unsigned long long)sh.nfp;
        } else {
            cnt->flags |= 4ULL;
        }
        sh.active = 0;
        sh.lazy_ns = defer ? ns : 0;
        *ctl = sh;
    }
}

// ---- wide kernels around the single-workgroup resolve (grid mode) ---------------------------------------------------------
// validation of the first round: one thread per history entry, spread over the chip (the probes are scattered reads,
// and one CU sustains only ~85 outstanding misses per microsecond)
__global__ __launch_bounds__(64) void k_validate(rs_args A)
{
    rs_shared *ctl = (rs_shared *)A.W.ctl;
    if (!ctl->active || ctl->ovf) return;
    const int nh = ctl->nhist < A.W.max_hist ? ctl->nhist : A.W.max_hist;
    const int ns = ctl->nslots0;        // slots that existed when the labels were written
    const double cr2i = A.P.collision_range * A.P.collision_range * AMC_CR2_INFLATE;
    for (int h = ctl->hist_begin + blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += gridDim.x * blockDim.x)
        rs_probe(A, A.G, ctl, A.W.sl_label, ns, A.W.max_slots, h, cr2i);
}

// commit: scratch state -> particle arrays, completed paths -> histogram / records, counters; clears the overlay
__global__ __launch_bounds__(256) void k_commit(rs_args A)
{
    const amc_resolve_ws &W = A.W;
    rs_shared *ctl = (rs_shared *)W.ctl;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
    if (A.apply_only) {
        // amc_flush: deferred results -> particle arrays (single block; the host clears its pending flag)
        const int ns = ctl->lazy_ns < W.max_slots ? ctl->lazy_ns : W.max_slots;
        for (int s = gtid; s < ns; s += gstride) {
            const int p = W.sl_p[s];
            if (W.slot_of[p] != s) continue;            // consumed by a streaming pass already
            if (W.sl_moved[s]) {
                A.S.x[p] = W.sl_x[s]; A.S.y[p] = W.sl_y[s]; A.S.z[p] = W.sl_z[s];
                A.S.vx[p] = W.sl_vx[s]; A.S.vy[p] = W.sl_vy[s]; A.S.vz[p] = W.sl_vz[s];
                A.S.d[p] = W.sl_d[s]; A.S.dx[p] = W.sl_dx[s]; A.S.dy[p] = W.sl_dy[s]; A.S.dz[p] = W.sl_dz[s];
                A.S.flag[p] = W.sl_flag[s];
            }
            W.slot_of[p] = -1;
        }
        return;
    }
    if (!ctl->active) return;
    const bool ok = ctl->ok && !ctl->ovf;
    const bool defer = ok && A.defer_commit;
    const int ns = ctl->nslots < W.max_slots ? ctl->nslots : W.max_slots;
    const int nev = ctl->nev < W.max_events ? ctl->nev : W.max_events;
    const int nh = ctl->nhist < W.max_hist ? ctl->nhist : W.max_hist;
    int my_hits = 0;
    for (int s = gtid; s < ns; s += gstride) {
        const int p = W.sl_p[s];
        if (ok && A.count_pp) my_hits += W.sl_hits[s];
        if (defer) continue;
        if (ok && W.sl_moved[s]) {
            A.S.x[p] = W.sl_x[s]; A.S.y[p] = W.sl_y[s]; A.S.z[p] = W.sl_z[s];
            A.S.vx[p] = W.sl_vx[s]; A.S.vy[p] = W.sl_vy[s]; A.S.vz[p] = W.sl_vz[s];
            A.S.d[p] = W.sl_d[s]; A.S.dx[p] = W.sl_dx[s]; A.S.dy[p] = W.sl_dy[s]; A.S.dz[p] = W.sl_dz[s];
            A.S.flag[p] = W.sl_flag[s];
        }
        W.slot_of[p] = -1;
    }
    // one atomic per wave instead of one per slot on a single counter word
    for (int o = 32; o > 0; o >>= 1) my_hits += __shfl_down(my_hits, o, 64);
    if ((threadIdx.x & 63) == 0 && my_hits) atomicAdd(&A.O.cnt->n_pp, (unsigned long long)my_hits);
    if (ok)
        for (int e = gtid; e < nev; e += gstride) {
            if (W.ev_gen[e] != W.sl_gen[W.ev_slot[e]]) continue;          // event of an emulation that was redone since
            const int owner = W.ev_which[e] ? W.ev_i[e] : W.ev_j[e];      // the particle whose free path completed
            if (owner < A.lo || owner >= A.hi) continue;
            amc_emit(A.O, W.ev_phase[e], W.ev_cell[e], W.ev_i[e], W.ev_j[e], W.ev_which[e], W.ev_val[4 * e + 0],
                     W.ev_val[4 * e + 1], W.ev_val[4 * e + 2], W.ev_val[4 * e + 3]);
        }
    for (int h = gtid; h < nh; h += gstride) W.ov_head[rs_hist_cell(A, A.G, h)] = -1;   // overlay entries of this sweep
    if (gtid == 0) {
        amc_dev_counters *cnt = A.O.cnt;
        cnt->n_candidates += (unsigned long long)ctl->ncand;
        cnt->n_clusters += (unsigned long long)ctl->nclusters;
        cnt->n_rounds += (unsigned long long)ctl->rounds;
        if (ok) {
            if (A.count_pp) cnt->n_fp_errors += (unsigned long long)ctl->nfp;
        } else {
            cnt->flags |= 4ULL;
        }
        ctl->lazy_ns = defer ? ns : 0;
    }
}

// ---- multi-GPU helpers ------------------------------------------------------------------------------------------------------
// candidate state table from the particle arrays (single GPU: the detect kernel gathers inline)
__global__ __launch_bounds__(256) void k_gather_cst(rs_args A)
{
    const amc_resolve_ws &W = A.W;
    const int ncand = min((int)A.O.cnt->cand_count, W.max_cand);
    const size_t m = (size_t)W.max_cand;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ncand; k += gridDim.x * blockDim.x) {
        const int pp[2] = {W.cand_j[k], W.cand_i[k]};
        for (int w = 0; w < 2; w++) {
            const int p = pp[w];
            double *t = W.cst + (size_t)(11 * w) * m + k;
            t[0 * m] = A.S.x[p]; t[1 * m] = A.S.y[p]; t[2 * m] = A.S.z[p];
            t[3 * m] = A.S.vx[p]; t[4 * m] = A.S.vy[p]; t[5 * m] = A.S.vz[p];
            t[6 * m] = A.S.d[p]; t[7 * m] = A.S.dx[p]; t[8 * m] = A.S.dy[p]; t[9 * m] = A.S.dz[p];
            t[10 * m] = A.S.flag[p] ? 1.0 : 0.0;
        }
    }
}

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

template <int GEOM>
static void rs_launch_all(amc_ctx *c, const rs_args &A)
{
    if (c->allpairs) {
        hipLaunchKernelGGL((k_resolve<GEOM, 2>), dim3(1), dim3(RS_T), 0, c->stream, A);
        return;
    }
    // launch plan from the candidate count of the most recent sweep the host has seen (a word the kernel writes into
    // host-mapped memory; no synchronisation, it may lag by a step): small sweeps need only this one kernel.  Either
    // plan is correct for any count — a wrong guess only costs time.
    if (c->h_host_ncand && *c->h_host_ncand <= RS_SMALL) {
        rs_args B = A;
        B.force_mono = 1;
        hipLaunchKernelGGL((k_resolve<GEOM, 0>), dim3(1), dim3(RS_T), 0, c->stream, B);
        return;
    }
    hipLaunchKernelGGL((k_resolve<GEOM, 0>), dim3(1), dim3(RS_T), 0, c->stream, A);
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_VALIDATE);
    hipLaunchKernelGGL(k_validate, dim3(128), dim3(64), 0, c->stream, A);
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_RESOLVE_MORE);
    hipLaunchKernelGGL((k_resolve<GEOM, 1>), dim3(1), dim3(RS_T), 0, c->stream, A);
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_COMMIT);
    hipLaunchKernelGGL(k_commit, dim3(64), dim3(256), 0, c->stream, A);
}

static rs_args rs_make_args(amc_ctx *c)
{
    rs_args A;
    A.P = c->P; A.S = c->S; A.G = c->G; A.B = c->B; A.W = c->W; A.O = c->out; A.n = c->n; A.allpairs = c->allpairs ? 1 : 0;
    A.dbg = c->d_dbg;
    A.single_round = 0;
    A.allow_mono = 1;
    A.force_mono = 0;
    A.host_ncand = c->d_host_ncand;
    A.defer_commit = 0;
    A.apply_only = 0;
    A.count_pp = c->mg_count_pp ? 1 : 0;
    A.lo = c->lo; A.hi = c->hi;
    A.inv_dx = c->P.dx > 0 ? 1.0 / c->P.dx : 0.0; A.inv_dy = c->P.dy > 0 ? 1.0 / c->P.dy : 0.0; A.inv_dz = c->P.dz > 0 ? 1.0 / c->P.dz : 0.0;
    return A;
}

hipError_t amc_launch_apply(amc_ctx *c)
{
    rs_args A = rs_make_args(c);
    A.apply_only = 1;
    hipLaunchKernelGGL(k_commit, dim3(8), dim3(256), 0, c->stream, A);
    return hipGetLastError();
}

hipError_t amc_launch_resolve(amc_ctx *c, bool defer_commit)
{
    rs_args A = rs_make_args(c);
    A.defer_commit = defer_commit ? 1 : 0;
    amc_prof_begin(c, AMC_K_RESOLVE);
    switch (c->P.geometry) {
    case AMC_GEOM_CELL: rs_launch_all<AMC_GEOM_CELL>(c, A); break;
    case AMC_GEOM_CUBE: rs_launch_all<AMC_GEOM_CUBE>(c, A); break;
    default: rs_launch_all<AMC_GEOM_PORE>(c, A); break;
    }
    amc_prof_end(c);
    return hipGetLastError();
}

// multi-GPU: one round (first: claim + round 1, else one continuation round), then the wide validation
template <int GEOM>
static void rs_launch_round(amc_ctx *c, rs_args A, int first)
{
    A.allow_mono = 0;           // the host drives the rounds (state exchange between them)
    if (first) {
        hipLaunchKernelGGL(k_gather_cst, dim3(64), dim3(256), 0, c->stream, A);
        hipLaunchKernelGGL((k_resolve<GEOM, 0>), dim3(1), dim3(RS_T), 0, c->stream, A);
    } else {
        A.single_round = 1;
        hipLaunchKernelGGL((k_resolve<GEOM, 1>), dim3(1), dim3(RS_T), 0, c->stream, A);
    }
    hipLaunchKernelGGL(k_validate, dim3(128), dim3(64), 0, c->stream, A);
}
hipError_t amc_launch_resolve_round(amc_ctx *c, int first)
{
    const rs_args A = rs_make_args(c);
    amc_prof_begin(c, AMC_K_RESOLVE);
    switch (c->P.geometry) {
    case AMC_GEOM_CUBE: rs_launch_round<AMC_GEOM_CUBE>(c, A, first); break;
    default: rs_launch_round<AMC_GEOM_PORE>(c, A, first); break;
    }
    amc_prof_end(c);
    return hipGetLastError();
}
hipError_t amc_launch_commit(amc_ctx *c)
{
    const rs_args A = rs_make_args(c);
    amc_prof_begin(c, AMC_K_RESOLVE);
    hipLaunchKernelGGL(k_commit, dim3(64), dim3(256), 0, c->stream, A);
    amc_prof_end(c);
    return hipGetLastError();
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
