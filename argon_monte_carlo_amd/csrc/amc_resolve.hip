// amc_resolve.hip — the ordered half of the p-p sweep: turns the candidate pairs into exactly the collisions the
// reference's sequential, in-place cell loops would perform (Pore:160-255 driven by Pore:520-549, Cube:231-336).
//
// Why this is not a plain parallel-for: the reference updates positions/velocities in place while it scans, cells
// overlap (a pair can be visited in up to 8 colour groups / cells) and a particle hit twice in one sweep sees the
// first collision's result.  Collisions are rare (~N/2000 per step) and almost always isolated,
// so the work is organised as OPTIMISTIC CLUSTERS WITH CONSERVATIVE VALIDATION:
//
//   1. clusters = connected components of the candidate graph (label propagation, label = lowest particle index);
//   2. one thread per cluster runs a LITERAL emulation of the reference restricted to the cluster's members —
//      same colour-group / cell order, same membership predicates evaluated at the same moments (Pore:527-530;
//      Cube's in_x/in_y/in_z masks are evaluated per x-layer / (x,y)-layer / cell, Cube:233-238), same i>j loop
//      order (members sorted by particle index), same arithmetic (amc_collide) — on a scratch copy of the state;
//   3. validation: every position a member occupied during the emulation is checked against all particles
//      outside its cluster (pre-sweep positions via the detection grid, other clusters' new positions via an
//      overlay list per grid cell).  Anything within collision_range*(1+1e-9) merges the clusters / pulls the
//      particle in, and the round is repeated from the untouched pre-sweep state.  When a round validates, no
//      test the reference performs between a member and a non-member can hit, hence the restricted emulation is
//      exactly what the reference computes;
//   4. commit: scratch state -> particle arrays, completed paths -> histogram / record buffer, counters.
//
// The whole kernel is ONE 1024-thread workgroup (phases separated by __syncthreads, counters in LDS): the data set
// is a few hundred pairs, so the cost is latency, not throughput.
#include "amc_grid_dev.h"

#define RS_T 1024
#define RS_SORT_LDS 8192
#define RS_MAX_ROUNDS 256
#define AMC_CR2_INFLATE (1.0 + 1.0e-9)

struct rs_args {
    amc_params P;
    amc_state S;
    amc_grid G;
    amc_sorted B;
    amc_resolve_ws W;
    amc_out O;
    long long n;
    int allpairs;
};

AMC_DEV amc_particle rs_load_slot(const amc_resolve_ws &W, int s)
{
    amc_particle q;
    q.x = W.sl_x[s]; q.y = W.sl_y[s]; q.z = W.sl_z[s]; q.vx = W.sl_vx[s]; q.vy = W.sl_vy[s]; q.vz = W.sl_vz[s];
    q.d = W.sl_d[s]; q.dx = W.sl_dx[s]; q.dy = W.sl_dy[s]; q.dz = W.sl_dz[s]; q.flag = W.sl_flag[s] != 0;
    return q;
}
AMC_DEV void rs_store_slot(const amc_resolve_ws &W, int s, const amc_particle &q)
{
    W.sl_x[s] = q.x; W.sl_y[s] = q.y; W.sl_z[s] = q.z; W.sl_vx[s] = q.vx; W.sl_vy[s] = q.vy; W.sl_vz[s] = q.vz;
    W.sl_d[s] = q.d; W.sl_dx[s] = q.dx; W.sl_dy[s] = q.dy; W.sl_dz[s] = q.dz; W.sl_flag[s] = q.flag ? 1 : 0;
}

struct rs_shared {
    int nslots, nedges, nhist, nev, dirty, changed, nhits, nfp, ovf, nclusters;
};

// one tested pair inside the emulation: exact overlap test on the scratch state, resolve on hit
AMC_DEV void rs_test_pair(const rs_args &A, rs_shared *sh, int sj, int si, int phase, long long cell)
{
    const amc_resolve_ws &W = A.W;
    if (!amc_overlap(W.sl_x[sj], W.sl_y[sj], W.sl_z[sj], W.sl_x[si], W.sl_y[si], W.sl_z[si], A.P.collision_range)) return;
    amc_particle p1 = rs_load_slot(W, sj), p2 = rs_load_slot(W, si);
    const int pi = W.sl_p[si], pj = W.sl_p[sj];
    auto emit = [&](int which, double tot, double px, double py, double pz) {
        const int e = atomicAdd(&sh->nev, 1);
        if (e < W.max_events) {
            W.ev_phase[e] = phase; W.ev_cell[e] = cell; W.ev_i[e] = pi; W.ev_j[e] = pj; W.ev_which[e] = which;
            W.ev_val[4 * e + 0] = tot; W.ev_val[4 * e + 1] = px; W.ev_val[4 * e + 2] = py; W.ev_val[4 * e + 3] = pz;
        } else {
            sh->ovf = 1;
        }
    };
    if (amc_collide(p1, p2, A.P.collision_range, A.P.argon_mass, emit)) {
        atomicAdd(&sh->nfp, 1);     // the reference would raise FloatingPointError here (Pore:11,185)
        return;
    }
    rs_store_slot(W, sj, p1);
    rs_store_slot(W, si, p2);
    W.sl_moved[sj] = 1; W.sl_moved[si] = 1;
    atomicAdd(&sh->nhits, 1);
    const int h = atomicAdd(&sh->nhist, 2);
    if (h + 1 < W.max_hist) {
        W.hist_slot[h] = sj; W.hist_x[h] = p1.x; W.hist_y[h] = p1.y; W.hist_z[h] = p1.z;
        W.hist_slot[h + 1] = si; W.hist_x[h + 1] = p2.x; W.hist_y[h + 1] = p2.y; W.hist_z[h + 1] = p2.z;
    } else {
        sh->ovf = 1;
    }
}

// literal emulation of the reference restricted to the members order[b..e) (sorted by particle index)
AMC_DEV void rs_emulate(const rs_args &A, rs_shared *sh, int b, int e)
{
    const amc_resolve_ws &W = A.W;
    const amc_params &P = A.P;
    const int *ord = W.order;
    if (P.geometry == AMC_GEOM_CELL) {
        // one cell holding everything: Pore:168-169 loop order
        for (int a = b + 1; a < e; a++)
            for (int c = b; c < a; c++) rs_test_pair(A, sh, ord[c], ord[a], 16, 0);
    } else if (P.geometry == AMC_GEOM_CUBE) {
        // Cube:231-336 — masks are evaluated where the reference evaluates them
        for (int lx = 0; lx < P.nx; lx++) {
            const double xlo = lx * P.dx - P.overlap_x, xhi = (lx + 1) * P.dx;               // Cube:233
            int cnt = 0;
            for (int a = b; a < e; a++) {
                const double v = W.sl_x[ord[a]];
                const int in = (xlo < v) && (v < xhi);
                W.sl_tmp[ord[a]] = in;
                cnt += in;
            }
            if (cnt < 2) continue;
            for (int ly = 0; ly < P.ny; ly++) {
                const double ylo = ly * P.dy - P.overlap_y, yhi = (ly + 1) * P.dy;           // Cube:235
                cnt = 0;
                for (int a = b; a < e; a++) {
                    const int s = ord[a];
                    int t = W.sl_tmp[s] & 1;
                    if (t) {
                        const double v = W.sl_y[s];
                        if ((ylo < v) && (v < yhi)) { t |= 2; cnt++; }
                    }
                    W.sl_tmp[s] = t;
                }
                if (cnt < 2) continue;
                for (int lz = 0; lz < P.nz; lz++) {
                    const double zlo = lz * P.dz - P.overlap_z, zhi = (lz + 1) * P.dz;       // Cube:237
                    cnt = 0;
                    for (int a = b; a < e; a++) {
                        const int s = ord[a];
                        int t = W.sl_tmp[s] & 3;
                        if (t == 3) {
                            const double v = W.sl_z[s];
                            if ((zlo < v) && (v < zhi)) { t |= 4; cnt++; }
                        }
                        W.sl_tmp[s] = t;
                    }
                    if (cnt < 2) continue;
                    const long long cell = ((long long)lx * P.ny + ly) * P.nz + lz;
                    for (int a = b + 1; a < e; a++) {
                        if (W.sl_tmp[ord[a]] != 7) continue;
                        for (int c = b; c < a; c++)
                            if (W.sl_tmp[ord[c]] == 7) rs_test_pair(A, sh, ord[c], ord[a], 16, cell);
                    }
                }
            }
        }
    } else {
        // Pore:520-549 == Temp:813-842 — 8 colour groups, membership from the positions at the start of the group
        const int nzl = P.nz / 2;
        for (int g = 0; g < 8; g++) {
            const int gx = g >> 2, gy = (g >> 1) & 1, gz = g & 1;                            // Pore:522-524
            int cnt = 0;
            for (int a = b; a < e; a++) {
                const int s = ord[a];
                int cell = -1;
                const int lx = amc_axis_cell(W.sl_x[s], gx, P.nx, P.nx, P.dx, P.overlap_x);
                if (lx >= 0) {
                    const int ly = amc_axis_cell(W.sl_y[s], gy, P.ny, P.ny, P.dy, P.overlap_y);
                    if (ly >= 0) {
                        const int lz = amc_axis_cell(W.sl_z[s], gz, nzl, 0, P.dz, P.overlap_z);
                        if (lz >= 0) cell = (lx * P.ny + ly) * nzl + lz;                     // Pore:530 list order
                    }
                }
                W.sl_tmp[s] = cell;
                cnt += cell >= 0;
            }
            if (cnt < 2) continue;
            for (int a = b + 1; a < e; a++) {
                const int ca = W.sl_tmp[ord[a]];
                if (ca < 0) continue;
                for (int c = b; c < a; c++)
                    if (W.sl_tmp[ord[c]] == ca) rs_test_pair(A, sh, ord[c], ord[a], 16 + g, ca);
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

// get (or create) the slot of particle p; creation is published by the barrier that follows the phase
AMC_DEV void rs_claim_slot(const amc_resolve_ws &W, rs_shared *sh, int p)
{
    const int old = atomicCAS(&W.slot_of[p], -1, -2);
    if (old == -1) {
        const int s = atomicAdd(&sh->nslots, 1);
        if (s < W.max_slots) {
            W.sl_p[s] = p;
            W.slot_of[p] = s;
        } else {
            sh->ovf = 1;
            W.slot_of[p] = -1;
        }
    }
}

AMC_DEV void rs_add_edge(const amc_resolve_ws &W, rs_shared *sh, int pa, int pb)
{
    const int k = atomicAdd(&sh->nedges, 1);
    if (k < W.max_edges) { W.edge_a[k] = pa; W.edge_b[k] = pb; } else sh->ovf = 1;
    sh->dirty = 1;
}

__global__ __launch_bounds__(RS_T) void k_resolve(rs_args A)
{
    __shared__ rs_shared sh;
    __shared__ unsigned long long lds_keys[RS_SORT_LDS];
    const amc_resolve_ws &W = A.W;
    const int tid = threadIdx.x;
    amc_dev_counters *cnt = A.O.cnt;
    int ncand = (int)cnt->cand_count;
    if (ncand > W.max_cand) ncand = W.max_cand;
    if (tid == 0) {
        sh.nslots = 0; sh.nedges = 0; sh.nhist = 0; sh.nev = 0; sh.dirty = 0; sh.changed = 0; sh.nhits = 0; sh.nfp = 0;
        sh.ovf = 0; sh.nclusters = 0;
    }
    __syncthreads();
    if (ncand == 0) return;     // uniform: nothing to resolve this sweep

    const double cr2i = A.P.collision_range * A.P.collision_range * AMC_CR2_INFLATE;

    // ---- slots for the candidate endpoints ----------------------------------------------------------------------
    for (int k = tid; k < ncand; k += RS_T) {
        rs_claim_slot(W, &sh, W.cand_i[k]);
        rs_claim_slot(W, &sh, W.cand_j[k]);
    }
    __syncthreads();

    int rounds = 0;
    for (;;) {
        rounds++;
        const int ns = sh.nslots < W.max_slots ? sh.nslots : W.max_slots;
        const int nedges = sh.nedges < W.max_edges ? sh.nedges : W.max_edges;
        __syncthreads();
        // ---- (re)load the scratch state from the untouched particle arrays; labels = own particle index --------
        for (int s = tid; s < ns; s += RS_T) {
            const int p = W.sl_p[s];
            W.sl_x[s] = A.S.x[p]; W.sl_y[s] = A.S.y[p]; W.sl_z[s] = A.S.z[p];
            W.sl_vx[s] = A.S.vx[p]; W.sl_vy[s] = A.S.vy[p]; W.sl_vz[s] = A.S.vz[p];
            W.sl_d[s] = A.S.d[p]; W.sl_dx[s] = A.S.dx[p]; W.sl_dy[s] = A.S.dy[p]; W.sl_dz[s] = A.S.dz[p];
            W.sl_flag[s] = A.S.flag[p];
            W.sl_moved[s] = 0;
            W.sl_label[s] = p;
        }
        if (tid == 0) { sh.nhist = 0; sh.nev = 0; sh.nhits = 0; sh.nfp = 0; sh.dirty = 0; sh.nclusters = 0; }
        __syncthreads();
        // ---- connected components by label propagation (label = lowest particle index of the cluster) ------------
        for (;;) {
            if (tid == 0) sh.changed = 0;
            __syncthreads();
            for (int k = tid; k < ncand + nedges; k += RS_T) {
                const int pa = k < ncand ? W.cand_i[k] : W.edge_a[k - ncand];
                const int pb = k < ncand ? W.cand_j[k] : W.edge_b[k - ncand];
                const int sa = W.slot_of[pa], sb = W.slot_of[pb];
                if (sa < 0 || sb < 0) continue;
                const int la = W.sl_label[sa], lb = W.sl_label[sb];
                if (la < lb) { atomicMin(&W.sl_label[sb], la); sh.changed = 1; }
                else if (lb < la) { atomicMin(&W.sl_label[sa], lb); sh.changed = 1; }
            }
            __syncthreads();
            if (!sh.changed) break;
            __syncthreads();
        }
        // ---- order the slots by (label, particle index) -------------------------------------------------------------
        int m = 1;
        while (m < ns) m <<= 1;
        unsigned long long *keys = (m <= RS_SORT_LDS) ? lds_keys : W.sl_key;
        for (int s = tid; s < m; s += RS_T)
            keys[s] = (s < ns) ? (((unsigned long long)(unsigned)W.sl_label[s] << 32) | (unsigned)W.sl_p[s]) : ~0ULL;
        __syncthreads();
        rs_bitonic(keys, m);
        for (int r = tid; r < ns; r += RS_T) W.order[r] = W.slot_of[(int)(keys[r] & 0xffffffffULL)];
        __syncthreads();
        // ---- emulate every cluster -------------------------------------------------------------------------------
        for (int r = tid; r < ns; r += RS_T) {
            const int lab = W.sl_label[W.order[r]];
            if (r > 0 && W.sl_label[W.order[r - 1]] == lab) continue;     // not a cluster head
            int e = r + 1;
            while (e < ns && W.sl_label[W.order[e]] == lab) e++;
            atomicAdd(&sh.nclusters, 1);
            if (e - r >= 2) rs_emulate(A, &sh, r, e);
        }
        __syncthreads();
        // ---- validate ------------------------------------------------------------------------------------------------
        const int nh = sh.nhist < W.max_hist ? sh.nhist : W.max_hist;
        if (!A.allpairs) {
            // overlay: history entries hashed into their grid cell
            for (int h = tid; h < nh; h += RS_T) {
                int cx, cy, cz;
                amc_grid_coords(A.G, W.hist_x[h], W.hist_y[h], W.hist_z[h], cx, cy, cz);
                const int c = amc_grid_cell(A.G, cx, cy, cz, nullptr);
                W.ov_next[h] = atomicExch(&W.ov_head[c], h);
            }
            __syncthreads();
            for (int h = tid; h < nh; h += RS_T) {
                const int sme = W.hist_slot[h];
                const int pme = W.sl_p[sme];
                const int lme = W.sl_label[sme];
                const double x = W.hist_x[h], y = W.hist_y[h], z = W.hist_z[h];
                int cx, cy, cz;
                amc_grid_coords(A.G, x, y, z, cx, cy, cz);
                for (int dz = -1; dz <= 1; dz++)
                    for (int dy = -1; dy <= 1; dy++) {
                        int c_lo, c_hi;
                        if (!amc_grid_row(A.G, cx, cy + dy, cz + dz, c_lo, c_hi)) continue;
                        // pre-sweep positions of every particle stored in these cells
                        const int q1 = A.B.cell_start[c_hi + 1];
                        for (int q = A.B.cell_start[c_lo]; q < q1; q++) {
                            const int idx = A.B.sidx[q];
                            if (idx == pme) continue;
                            const double ex = A.B.sx[q] - x, ey = A.B.sy[q] - y, ez = A.B.sz[q] - z;
                            if (ex * ex + ey * ey + ez * ez < cr2i) {
                                const int so = W.slot_of[idx];
                                if (so >= 0 && so < ns && W.sl_label[so] == lme) continue;
                                if (so < 0) rs_claim_slot(W, &sh, idx);
                                rs_add_edge(W, &sh, pme, idx);
                            }
                        }
                        // new positions of other clusters' members
                        for (int c = c_lo; c <= c_hi; c++)
                            for (int h2 = W.ov_head[c]; h2 >= 0; h2 = W.ov_next[h2]) {
                                const int s2 = W.hist_slot[h2];
                                if (W.sl_label[s2] == lme) continue;
                                const double ex = W.hist_x[h2] - x, ey = W.hist_y[h2] - y, ez = W.hist_z[h2] - z;
                                if (ex * ex + ey * ey + ez * ez < cr2i) rs_add_edge(W, &sh, pme, W.sl_p[s2]);
                            }
                    }
            }
            __syncthreads();
            for (int h = tid; h < nh; h += RS_T) {
                int cx, cy, cz;
                amc_grid_coords(A.G, W.hist_x[h], W.hist_y[h], W.hist_z[h], cx, cy, cz);
                W.ov_head[amc_grid_cell(A.G, cx, cy, cz, nullptr)] = -1;
            }
        } else {
            // no grid (single cell / small N): brute force against all particles and all history entries
            for (long long w = tid; w < (long long)nh * A.n; w += RS_T) {
                const int h = (int)(w / A.n);
                const int idx = (int)(w % A.n);
                const int sme = W.hist_slot[h];
                if (idx == W.sl_p[sme]) continue;
                const double ex = A.S.x[idx] - W.hist_x[h], ey = A.S.y[idx] - W.hist_y[h], ez = A.S.z[idx] - W.hist_z[h];
                if (ex * ex + ey * ey + ez * ez < cr2i) {
                    const int so = W.slot_of[idx];
                    if (so >= 0 && so < ns && W.sl_label[so] == W.sl_label[sme]) continue;
                    if (so < 0) rs_claim_slot(W, &sh, idx);
                    rs_add_edge(W, &sh, W.sl_p[sme], idx);
                }
            }
            for (long long w = tid; w < (long long)nh * nh; w += RS_T) {
                const int h = (int)(w / nh), h2 = (int)(w % nh);
                if (h2 >= h) continue;
                const int s1 = W.hist_slot[h], s2 = W.hist_slot[h2];
                if (W.sl_label[s1] == W.sl_label[s2]) continue;
                const double ex = W.hist_x[h2] - W.hist_x[h], ey = W.hist_y[h2] - W.hist_y[h], ez = W.hist_z[h2] - W.hist_z[h];
                if (ex * ex + ey * ey + ez * ez < cr2i) rs_add_edge(W, &sh, W.sl_p[s1], W.sl_p[s2]);
            }
        }
        __syncthreads();
        if (!sh.dirty || sh.ovf || rounds >= RS_MAX_ROUNDS) break;
        __syncthreads();
    }

    // ---- commit --------------------------------------------------------------------------------------------------
    const int ns = sh.nslots < W.max_slots ? sh.nslots : W.max_slots;
    const bool ok = !sh.ovf && !(sh.dirty && rounds >= RS_MAX_ROUNDS);
    for (int s = tid; s < ns; s += RS_T) {
        const int p = W.sl_p[s];
        if (ok && W.sl_moved[s]) {
            A.S.x[p] = W.sl_x[s]; A.S.y[p] = W.sl_y[s]; A.S.z[p] = W.sl_z[s];
            A.S.vx[p] = W.sl_vx[s]; A.S.vy[p] = W.sl_vy[s]; A.S.vz[p] = W.sl_vz[s];
            A.S.d[p] = W.sl_d[s]; A.S.dx[p] = W.sl_dx[s]; A.S.dy[p] = W.sl_dy[s]; A.S.dz[p] = W.sl_dz[s];
            A.S.flag[p] = W.sl_flag[s];
        }
        W.slot_of[p] = -1;
    }
    if (ok) {
        const int nev = sh.nev < W.max_events ? sh.nev : W.max_events;
        for (int e = tid; e < nev; e += RS_T)
            amc_emit(A.O, W.ev_phase[e], W.ev_cell[e], W.ev_i[e], W.ev_j[e], W.ev_which[e], W.ev_val[4 * e + 0],
                     W.ev_val[4 * e + 1], W.ev_val[4 * e + 2], W.ev_val[4 * e + 3]);
    }
    __syncthreads();
    if (tid == 0) {
        cnt->n_candidates += (unsigned long long)ncand;
        cnt->n_clusters += (unsigned long long)sh.nclusters;
        cnt->n_rounds += (unsigned long long)rounds;
        if (ok) {
            cnt->n_pp += (unsigned long long)sh.nhits;
            cnt->n_fp_errors += (unsigned long long)sh.nfp;
        } else {
            cnt->flags |= 4ULL;
        }
        cnt->cand_count = 0;
    }
}

hipError_t amc_launch_resolve(amc_ctx *c)
{
    rs_args A;
    A.P = c->P; A.S = c->S; A.G = c->G; A.B = c->B; A.W = c->W; A.O = c->out; A.n = c->n; A.allpairs = c->allpairs ? 1 : 0;
    amc_prof_begin(c, AMC_K_RESOLVE);
    hipLaunchKernelGGL(k_resolve, dim3(1), dim3(RS_T), 0, c->stream, A);
    amc_prof_end(c);
    return hipGetLastError();
}
