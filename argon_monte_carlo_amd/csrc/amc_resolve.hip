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
// is kept off that single CU: detect gathers the candidates' state into a SoA table (coalesced reads here), and for
// large sweeps (> 430 candidates) the isolated pairs are emulated by a wide kernel first (k_pairs_wide), the first
// round's validation probes and the commit scatter are separate wide kernels, and only the entangled remainder and
// the later rounds stay in the workgroup.  A multi-particle cluster is emulated by a whole wave (rs_emulate_coop).
#include "amc_resolve_dev.h"

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

// MODE 0: first round only (claim, label, emulate), validation + commit are the wide kernels that follow
// MODE 1: continuation: if the wide validation found merges, run the remaining rounds (validation in-kernel)
// MODE 2: everything in one kernel incl. brute-force validation and commit (no detection grid: single cells, small N)
template <int GEOM, int MODE>
__global__ __launch_bounds__(RS_T) void k_resolve(rs_args A)
{
    long long t_last = (A.dbg && threadIdx.x == 0) ? wall_clock64() : 0;
    __shared__ rs_shared sh;
    __shared__ int wide_ns;             // slots made by the wide pair kernel in this sweep, -1 if it did not run
    __shared__ int wide_nh;             // history entries it made (already in the overlay lists)
    __shared__ int s_heads[64], s_nheads;   // first member of every multi-particle cluster when all waves share them
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
    if (threadIdx.x == 0) { wide_ns = -1; wide_nh = 0; }
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
            // what the wide pair kernel did before this one (slots, history, events of the isolated pairs); zero if it
            // did not run.  Its control block is re-armed for the next sweep.
            sh.nslots = 0; sh.nhist = 0; sh.nev = 0; sh.nfp = 0; sh.ovf = 0;
            if (A.wide_plan) {
                rs_shared *wc = (rs_shared *)W.wctl;
                sh.nslots = wc->nslots < W.max_slots ? wc->nslots : W.max_slots;
                sh.nhist = wc->nhist; sh.nev = wc->nev; sh.nfp = wc->nfp; sh.ovf = wc->ovf;
                wide_ns = wc->active ? sh.nslots : -1;             // (>= 0: the wide kernel ran, W.cand_done is valid)
                wide_nh = sh.nhist < W.max_hist ? sh.nhist : W.max_hist;
                wc->nslots = 0; wc->nhist = 0; wc->nev = 0; wc->nfp = 0; wc->ovf = 0; wc->active = 0; wc->cur_round = 1;
            }
            sh.nslots0 = 0;
            sh.nedges = 0; sh.dirty = 0; sh.changed = 0; sh.nhits = 0;
            sh.nclusters = 0; sh.ncomplex = 0; sh.rounds = 0; sh.ncand = ncand;
            sh.active = ncand > 0; sh.ok = 1; sh.edges_done = 0; sh.hist_begin = 0; sh.cur_round = 0;
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
    const bool mono = (MODE == 2) || (MODE == 0 && (A.force_mono || ncand <= A.plan_small));
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
        const bool wide = wide_ns >= 0;
        for (int k = tid; k < ncand; k += RS_T) {
            if (wide && W.cand_done[k]) continue;           // isolated pair: slots, state and history exist already
            rs_claim_slot(W, &sh, V.cap, W.cand_i[k]);
            rs_claim_slot(W, &sh, V.cap, W.cand_j[k]);
        }
        __syncthreads();
        for (int k = tid; k < ncand; k += RS_T) {
            if (wide && W.cand_done[k]) continue;
            W.cand_si[k] = W.slot_of[W.cand_i[k]];
            W.cand_sj[k] = W.slot_of[W.cand_j[k]];
        }
        __syncthreads();
    }
    RS_STAMP(0);

    int rounds = sh.rounds;
    int edges_done = sh.edges_done;     // edges [0, edges_done) already hold slot ids
    int ns_lab = 0;                     // slots that carry a label of the previous round
    if (MODE == 1) {
        // continuation: the labels of the round(s) before are in global memory (written at the hand-over)
        ns_lab = sh.nslots0 < V.cap ? sh.nslots0 : V.cap;
        if (V.label != W.sl_label)
            for (int s = tid; s < ns_lab; s += RS_T) V.label[s] = W.sl_label[s];
        __syncthreads();
    }
    for (;;) {
        rounds++;
        const int ns = sh.nslots < V.cap ? sh.nslots : V.cap;
        const int nedges = sh.nedges < W.max_edges ? sh.nedges : W.max_edges;
        __syncthreads();
        // ---- per-round reset; new merge edges: particle ids -> slot ids -------------------------------------------------
        const bool first = (rounds == 1);
        for (int s = tid; s < ns; s += RS_T) {
            if (first || s >= ns_lab) V.label[s] = s;       // later rounds keep the previous round's labels
            V.size[s] = first ? 0 : s;                      // (later rounds: parent map of the cluster merges, see below)
            vdirty[s] = first;                              // round 1 emulates everything; later rounds only what the new edges touch
        }
        for (int k = edges_done + tid; k < nedges; k += RS_T) {
            W.edge_a[k] = W.slot_of[W.edge_a[k]];
            W.edge_b[k] = W.slot_of[W.edge_b[k]];
        }
        const int edges_new = edges_done;
        edges_done = nedges;
        if (tid == 0) { s_nheads = 0; sh.dirty = 0; sh.nclusters = 0; sh.ncomplex = 0; sh.cur_round = rounds; sh.hist_begin = first ? 0 : (sh.nhist < W.max_hist ? sh.nhist : W.max_hist); }   // (round 1 also validates what the wide pair kernel produced)
        __syncthreads();
        if (first) {
            // ---- connected components by label propagation (label = lowest slot id of the cluster); the pairs the wide
            // kernel emulated are isolated by construction: their label is known without propagation ----------------------
            const bool wide1 = wide_ns >= 0;
            if (wide1)
                for (int k = tid; k < ncand; k += RS_T)
                    if (W.cand_done[k]) V.label[W.cand_si[k]] = W.cand_sj[k];              // (sj = si - 1)
            for (;;) {
                int changed = 0;
                for (int k = tid; k < ncand; k += RS_T) {
                    if (wide1 && W.cand_done[k]) continue;
                    const int sa = W.cand_si[k], sb = W.cand_sj[k];
                    if (sa < 0 || sb < 0 || sa >= ns || sb >= ns) continue;
                    const int la = V.label[sa], lb = V.label[sb];
                    if (la < lb) { atomicMin(&V.label[sb], la); changed = 1; }
                    else if (lb < la) { atomicMin(&V.label[sa], lb); changed = 1; }
                }
                if (!__syncthreads_or(changed)) break;
            }
        } else {
            // ---- later rounds: the clusters of the previous round merge along the NEW edges only — a union over label
            // values (parent map in V.size, roots = minimum label) and one relabelling pass, instead of propagating
            // over every candidate again ----------------------------------------------------------------------------------------
            for (int k = edges_new + tid; k < nedges; k += RS_T) {
                const int sa = W.edge_a[k], sb = W.edge_b[k];
                if (sa < 0 || sb < 0 || sa >= ns || sb >= ns) continue;
                rs_union(V.size, V.label[sa], V.label[sb]);
            }
            __syncthreads();
            for (int s = tid; s < ns; s += RS_T) V.label[s] = rs_find(V.size, V.label[s]);
            __syncthreads();
            for (int s = tid; s < ns; s += RS_T) V.size[s] = 0;
            __syncthreads();
        }
        ns_lab = ns;
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
            if (vdirty[V.label[s]] && !(rounds == 1 && s < wide_ns)) { W.sl_moved[s] = 0; W.sl_gen[s] = rounds; W.sl_hits[s] = 0; }
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
            } else {
                // three or more clusters: every wave of the workgroup takes its share of them after the pair phase
                // (below) — each cluster by a whole wave again; this wave only publishes where the clusters start
                if (is_head) s_heads[__popcll(__ballot(is_head) & ((1ULL << w) - 1ULL))] = w;
                if (w == 0) s_nheads = nheads;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (A.dbg && tid == 0) { const long long n__ = wall_clock64(); A.dbg[3] += n__ - t_last; t_last = n__; }
            if (nheads <= 2 && w < nc && K.moved[w]) rs_store_slot(W, K.slot[w], rs_load_work(K, w));
        } else {
            // ---- two-particle clusters straight from the candidate list, both particles in registers ------------------------
            const int t0 = split ? tid - 64 : tid, tstride = split ? RS_T - 64 : RS_T;
            for (int k = t0; k < ncand; k += tstride) {
                const int si = W.cand_si[k], sj = W.cand_sj[k];
                if (si < 0 || sj < 0 || si >= ns || sj >= ns) continue;
                if (V.size[V.label[si]] != 2 || !vdirty[V.label[si]]) continue;
                if (rounds == 1 && wide_ns >= 0 && W.cand_done[k]) continue;       // emulated by the wide pair kernel
                rs_emulate_pair<GEOM>(A, &sh, k, W.cand_j[k], W.cand_i[k], sj, si);
            }
        }
        RS_STAMP(7);
        __syncthreads();
        if (split && s_nheads > 2) {
            const unsigned long long *sorted = lds_keys + RS_SORT_LDS / 2;
            for (int hx = tid >> 6; hx < s_nheads; hx += RS_T / 64) {           // wave-uniform: one cluster per wave and turn
                const int w0 = s_heads[hx];
                const unsigned lab = (unsigned)(sorted[w0] >> 32);
                int e = w0 + 1;
                while (e < nc && (unsigned)(sorted[e] >> 32) == lab) e++;
                if (e - w0 >= 2) {
                    if (e - w0 <= RS_COOP_MAX) rs_emulate_coop(A, &sh, K, w0, e);
                    else if ((tid & 63) == 0) rs_emulate_generic(A, &sh, K, w0, e);
                }
            }
            __syncthreads();
            if (tid < nc && K.moved[tid]) rs_store_slot(W, K.slot[tid], rs_load_work(K, tid));
            __syncthreads();
        }
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
        if (!mono && MODE == 0) {
            // ---- hand over to the wide validation kernel: labels to global memory, history into the overlay -------------
            if (V.label != W.sl_label)
                for (int s = tid; s < ns; s += RS_T) W.sl_label[s] = V.label[s];
            for (int h = (sh.hist_begin > wide_nh ? sh.hist_begin : wide_nh) + tid; h < nh; h += RS_T) W.ov_next[h] = atomicExch(&W.ov_head[rs_hist_cell(A, G, h)], h);
            __syncthreads();
            if (tid == 0) { sh.rounds = rounds; sh.edges_done = edges_done; sh.nslots0 = ns; *ctl = sh; }
            RS_STAMP(4);
            if (A.dbg && tid == 0) { A.dbg[8] += 1; A.dbg[9] += ncand; A.dbg[10] += sh.ncomplex; A.dbg[11] += 1; }
            return;
        }
        // ---- validate: every new position against everything outside its cluster ------------------------------------------------
        if (!A.allpairs) {
            for (int h = (sh.hist_begin > wide_nh ? sh.hist_begin : wide_nh) + tid; h < nh; h += RS_T) W.ov_next[h] = atomicExch(&W.ov_head[rs_hist_cell(A, G, h)], h);
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
            if (A.count_pp) {           // (multi-GPU: every rank resolves every collision, one of them counts)
                cnt->n_pp += (unsigned long long)sh.nhits;
                cnt->n_fp_errors += (unsigned long long)sh.nfp;
            }
        } else {
            cnt->flags |= 4ULL;
        }
        sh.active = 0;
        sh.lazy_ns = defer ? ns : 0;
        *ctl = sh;
    }
}

// ---- isolated pairs, wide -----------------------------------------------------------------------------------------------------
// Everything in k_resolve costs time in proportion to the number of candidates (~38 ns each on its single CU).  Nine
// candidates in ten are isolated pairs — both particles appear in no other candidate (degree one, counted by the detect
// kernel) — and need none of its machinery: no claim race, no labels, no ordering against other pairs.  This kernel
// gives them their slots (two per pair from one counter, allocated per wave) and emulates them exactly like the
// workgroup would (rs_emulate_pair); the workgroup then starts from the counters left in W.wctl and handles only the
// entangled remainder.  Validation treats both kinds alike (the pair's history entries are probed by k_validate).
template <int GEOM>
__global__ __launch_bounds__(256) void k_pairs_wide(rs_args A)
{
    const amc_resolve_ws &W = A.W;
    rs_shared *wc = (rs_shared *)W.wctl;
    int ncand = (int)A.O.cnt->cand_count;
    if (ncand > W.max_cand) ncand = W.max_cand;
    if (blockIdx.x == 0 && threadIdx.x == 0) wc->active = 1;
    const unsigned int one = (A.sweep_epoch << 2) | 1u;
    const int lane = threadIdx.x & 63, stride = gridDim.x * blockDim.x;
    for (int k0 = blockIdx.x * blockDim.x + (threadIdx.x - lane); k0 < ncand; k0 += stride) {     // wave-uniform trip count
        const int k = k0 + lane;
        const bool valid = k < ncand;
        int pi = 0, pj = 0;
        bool iso = false;
        if (valid) {
            pi = W.cand_i[k]; pj = W.cand_j[k];
            iso = W.deg[pi] == one && W.deg[pj] == one;
        }
        const int base = rs_count_add(&wc->nslots, iso ? 2 : 0);
        if (iso && base + 1 >= W.max_slots) { wc->ovf = 1; iso = false; }
        if (valid) W.cand_done[k] = iso ? 1 : 0;
        if (!iso) continue;
        const int sj = base, si = base + 1;
        W.slot_of[pj] = sj; W.slot_of[pi] = si;
        W.sl_p[sj] = pj; W.sl_p[si] = pi;
        W.cand_sj[k] = sj; W.cand_si[k] = si;
        W.sl_moved[sj] = 0; W.sl_moved[si] = 0;
        W.sl_gen[sj] = 1; W.sl_gen[si] = 1;
        atomicAnd(&W.sl_hits[sj], 0); atomicAnd(&W.sl_hits[si], 0);      // (atomics, like the increments that follow)
        rs_emulate_pair<GEOM>(A, wc, k, pj, pi, sj, si);
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
    if ((threadIdx.x & 63) == 0 && my_hits) atomicAdd(&A.O.banks[amc_bank_id()].n_pp, (unsigned long long)my_hits);
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
    if (!c->plan_split) {
        rs_args B = A;
        B.force_mono = 1;
        hipLaunchKernelGGL((k_resolve<GEOM, 0>), dim3(1), dim3(RS_T), 0, c->stream, B);
        return;
    }
    rs_args Aw = A;
    Aw.wide_plan = 1;
    amc_prof_cancel(c);                 // (the caller's bracket is re-opened below, around the kernel it is named after)
    amc_prof_begin(c, AMC_K_PAIRS_WIDE);
    hipLaunchKernelGGL((k_pairs_wide<GEOM>), dim3(64), dim3(256), 0, c->stream, Aw);
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_RESOLVE);
    hipLaunchKernelGGL((k_resolve<GEOM, 0>), dim3(1), dim3(RS_T), 0, c->stream, Aw);
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
    A.force_mono = 0;
    A.host_ncand = c->d_host_ncand;
    A.plan_small = c->plan_small;
    A.defer_commit = 0;
    A.apply_only = 0;
    A.sweep_epoch = c->sweep_epoch;
    A.wide_plan = 0;
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
