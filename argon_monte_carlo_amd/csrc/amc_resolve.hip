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
// Who does what (grid mode): steps 1-3 of the FIRST round run wide, over the whole chip, in k_clusters_wide
// (amc_clusters.hip) for every connected component of up to 16 particles — each validates its own new positions there.
// What is left for the ONE 512-thread workgroup of this file (phases separated by __syncthreads, counters in LDS; its
// cost is latency, not throughput) is the entangled remainder: components too large for the wide kernel (claim,
// label, emulate, validate as below), the later rounds after a validation found an outsider (merge along the new
// edges, re-emulate only the touched clusters from the untouched pre-sweep state, validate again).  When the wide
// kernel left nothing and found nothing — 98 % of the sweeps at N = 1e5 — this kernel only hands the counters over.  The
// commit is wide work again (amc_commit_dev.h): it rides along with the next streaming pass or runs as k_commit.
// Without a detection grid (single cells, N <= 4096: MODE 2) everything, brute-force validation and commit included,
// happens here.
#include <stdlib.h>
#include <algorithm>

#include "amc_resolve_dev.h"
#include "amc_commit_dev.h"

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

// MODE 0: grid mode, after k_clusters_wide (A.wide_plan): the remainder, the later rounds and — when the host launched
//         only this kernel after it (A.force_mono, small sweeps) — the commit; else k_commit follows
// MODE 2: no detection grid (single cells, small N): everything in one kernel incl. brute-force validation and commit
template <int GEOM, int MODE>
__global__ __launch_bounds__(RS_T) void k_resolve(rs_args A)
{
    long long t_last = (A.dbg && threadIdx.x == 0) ? wall_clock64() : 0;
    if (A.dbg && MODE == 0 && A.wide_plan) {        // span of the wide kernel's launch: first working wave in -> last one out
        __shared__ unsigned long long s_in, s_out;
        if (threadIdx.x == 0) { s_in = ~0ULL; s_out = 0ULL; }
        __syncthreads();
        for (int w = threadIdx.x; w < 512; w += RS_T) {
            const long long *R = A.dbg + 128 + 128 * (long long)w;
            if (R[2] == (long long)A.sweep_epoch) { atomicMin(&s_in, (unsigned long long)R[0]); atomicMax(&s_out, (unsigned long long)R[1]); }
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_out > 0) { A.dbg[27] += (long long)(s_out - s_in); A.dbg[26] += t_last - (long long)s_out; A.dbg[25] += 1; }
        __syncthreads();
    }
    __shared__ rs_shared sh;
    __shared__ int wide_ns;             // slots made by the wide cluster kernel in this sweep, -1 if it did not run
    __shared__ int wide_nh;             // history entries it made (published and validated there)
    __shared__ int wide_dirty;          // its validation found merge edges
    __shared__ int s_left;              // candidates it left to this kernel
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
    int ncand = (int)cnt->cand_count;
    if (ncand > W.max_cand) ncand = W.max_cand;
    if (tid == 0) {
        // what the wide cluster kernel did before this one (slots, history, events, merge edges); zero if it did not
        // run.  Its control block is re-armed for the next sweep.
        wide_ns = -1; wide_nh = 0; wide_dirty = 0; s_left = 0;
        sh.nslots = 0; sh.nhist = 0; sh.nev = 0; sh.nfp = 0; sh.ovf = 0; sh.nedges = 0; sh.nclusters = 0;
        if (MODE == 0 && A.wide_plan) {
            // (candidate k owns slots 2k, 2k + 1 and the history pairs 4k, 4k + 2; what the wide kernel took from its
            // counters comes after those)
            rs_shared *wc = (rs_shared *)W.wctl;
            const int off = 2 * ncand;
            sh.nslots = off + wc->nslots < W.max_slots ? off + wc->nslots : W.max_slots;
            sh.nhist = 2 * off + wc->nhist; sh.nfp = wc->nfp; sh.ovf = wc->ovf; sh.nedges = wc->nedges;
            sh.nclusters = 0;
            for (int b = 0; b < 16; b++) { sh.nclusters += W.wctl[32 + b]; W.wctl[32 + b] = 0; }      // (banked by the wide kernel's waves)
            wide_dirty = wc->dirty;
            wide_ns = wc->active ? sh.nslots : -1;              // (>= 0: the wide kernel ran, W.cand_s[k].z is valid)
            wide_nh = sh.nhist < W.max_hist ? sh.nhist : W.max_hist;
            s_left = wc->changed ? -1 : 0;                      // it left components too large for it (how many candidates: counted below)
            wc->nslots = 0; wc->nhist = 0; wc->nedges = 0; wc->nfp = 0; wc->ovf = 0; wc->dirty = 0; wc->nclusters = 0;
            wc->changed = 0; wc->active = 0; wc->cur_round = 1;
            if (wide_ns < 0) { sh.nslots = 0; sh.nhist = 0; s_left = ncand; }        // (it did not run)
        } else {
            s_left = ncand;
        }
        sh.nslots0 = 0;
        sh.dirty = 0; sh.changed = 0; sh.nhits = 0;
        sh.ncomplex = 0; sh.rounds = 0; sh.ncand = ncand;
        sh.active = ncand > 0; sh.ok = 1; sh.edges_done = 0; sh.hist_begin = 0; sh.cur_round = 0;
        sh.lazy_ns = 0;         // the streaming pass before this sweep consumed the previous sweep's deferred results
        cnt->cand_count = 0;
        if (A.host_ncand) *A.host_ncand = ncand;
        if (ncand == 0) *ctl = sh;
    }
    __syncthreads();
    if (ncand == 0) return;     // uniform: nothing to resolve this sweep

    const bool wide = wide_ns >= 0;
    const int gen_off = wide ? 16 : 0;          // (the wide kernel tags its own re-emulations of a cluster with rounds 1, 2, 3)
    const bool mono = (MODE == 2);      // commit in this kernel (no-grid mode); with a grid the commit is wide work of its own:
                                        // it rides along with the next streaming pass or runs as k_commit
    if (s_left < 0) {                   // (rare: the wide kernel met a component beyond its capacity)
        __syncthreads();
        if (tid == 0) s_left = 0;
        __syncthreads();
        int mine = 0;
        for (int k = tid; k < ncand; k += RS_T) mine += !W.cand_s[k].z;
        if (mine) atomicAdd(&s_left, mine);
        __syncthreads();
    }
    const int nleft = s_left;
    // nothing left and nothing found by the wide kernel's validation: the sweep is resolved, only the commit remains
    const bool resolved = wide && nleft == 0 && !wide_dirty;
    RS_STAMP(0);

    const double cr2i = A.P.collision_range * A.P.collision_range * AMC_CR2_INFLATE;
    rs_slots V;
    unsigned char *vdirty;
    // (the slot arrays in global memory hold W.max_slots entries: labels in LDS must not let validation claim more)
    if ((wide ? wide_ns : 0) + 2 * nleft + 256 <= RS_NS) { V.label = s_label; V.size = s_size; V.cap = RS_NS < W.max_slots ? RS_NS : W.max_slots; vdirty = s_dirty; }
    else { V.label = W.sl_label; V.size = W.sl_tmp; V.cap = W.max_slots; vdirty = W.sl_dirty; }
    amc_grid G = A.G;
    int rounds = 0;
    if (!resolved) {
        // grid layer tables -> LDS (every validation probe reads them)
        if (!A.allpairs && 3 * G.gz <= RS_LAY) {
            for (int k = tid; k < 3 * G.gz; k += RS_T) s_lay[k] = A.G.lay_lo[k];     // the three tables are contiguous
            G.lay_lo = s_lay; G.lay_n = s_lay + G.gz; G.lay_off = s_lay + 2 * G.gz;
        }
        // ---- slots for the endpoints of the candidates that are left; candidates become slot pairs ----------------------
        for (int k = tid; k < ncand; k += RS_T) {
            if (wide && W.cand_s[k].z) continue;            // its cluster was emulated wide: slots, state and history exist
            const int4 c4 = W.cand4[k];
            rs_claim_slot(W, &sh, V.cap, c4.x);
            rs_claim_slot(W, &sh, V.cap, c4.y);
        }
        __syncthreads();
        for (int k = tid; k < ncand; k += RS_T) {
            if (wide && W.cand_s[k].z) continue;
            const int4 c4 = W.cand4[k];
            W.cand_s[k] = make_int4(W.slot_of[c4.x], W.slot_of[c4.y], 0, 0);
        }
        __syncthreads();
    }
    RS_STAMP(1);

    int ns_lab = 0;                     // slots that carry a label of the previous round
    int edges_conv = 0;                 // edges [0, edges_conv) hold slot ids
    int edges_merged = 0;               // edges [0, edges_merged) have been merged into the labels
    while (!resolved) {
        rounds++;
        const int ns = sh.nslots < V.cap ? sh.nslots : V.cap;
        const int nedges = sh.nedges < W.max_edges ? sh.nedges : W.max_edges;
        __syncthreads();
        // ---- per-round reset; new merge edges: particle ids (or encoded slots) -> slot ids -----------------------------------
        const bool first = (rounds == 1);
        for (int s = tid; s < ns; s += RS_T) {
            if (first) {
                // slots of the wide kernel carry the labels it wrote (first slot of the cluster) and are NOT emulated
                // again in this round; everything claimed here starts as its own cluster and is
                V.label[s] = (s < wide_ns) ? W.sl_meta[s].y : s;
            } else if (s >= ns_lab) {
                V.label[s] = s;                             // claimed by the previous validation
            }
            V.size[s] = first ? 0 : s;                      // (later rounds: parent map of the cluster merges, see below)
            vdirty[s] = first && s >= wide_ns;              // later rounds: only what the new edges touch
        }
        for (int k = edges_conv + tid; k < nedges; k += RS_T) {
            const int ea = W.edge_a[k], eb = W.edge_b[k];
            const int sa = ea < -1 ? -(ea + 2) : W.slot_of[ea], sb = eb < -1 ? -(eb + 2) : W.slot_of[eb];
            W.edge_a[k] = sa;
            W.edge_b[k] = sb;
            if (sa < 0 || sb < 0) sh.ovf = 1;       // (an end without a slot cannot happen; fail the step rather than merge nothing forever)
        }
        edges_conv = nedges;
        if (tid == 0) { s_nheads = 0; sh.dirty = 0; sh.ncomplex = 0; sh.cur_round = rounds + gen_off; sh.hist_begin = first ? wide_nh : (sh.nhist < W.max_hist ? sh.nhist : W.max_hist); }
        __syncthreads();
        if (first) {
            // ---- connected components of what is left by label propagation (label = lowest slot id of the cluster) ---------------
            for (;;) {
                int changed = 0;
                for (int k = tid; k < ncand; k += RS_T) {
                    const int4 cs = W.cand_s[k];
                    if (wide && cs.z) continue;
                    const int sa = cs.x, sb = cs.y;
                    if (sa < 0 || sb < 0 || sa >= ns || sb >= ns) continue;
                    const int la = V.label[sa], lb = V.label[sb];
                    if (la < lb) { atomicMin(&V.label[sb], la); changed = 1; }
                    else if (lb < la) { atomicMin(&V.label[sa], lb); changed = 1; }
                }
                if (!__syncthreads_or(changed)) break;
            }
        } else {
            // ---- later rounds: the clusters merge along the NEW edges only — a union over label values (parent map in
            // V.size, roots = minimum label) and one relabelling pass, instead of propagating over every candidate again ----
            for (int k = edges_merged + tid; k < nedges; k += RS_T) {
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
        // ---- cluster sizes, accumulated on the label slot ------------------------------------------------------------------
        for (int s = tid; s < ns; s += RS_T) atomicAdd(&V.size[V.label[s]], 1);
        // clusters touched by the merge edges of the previous validation are re-emulated; everything else keeps its
        // results, events and history (tagged with the round they were produced in)
        if (!first)
            for (int k = edges_merged + tid; k < nedges; k += RS_T) {
                const int sa = W.edge_a[k], sb = W.edge_b[k];
                if (sa >= 0 && sa < ns) vdirty[V.label[sa]] = 1;
                if (sb >= 0 && sb < ns) vdirty[V.label[sb]] = 1;
            }
        if (!first) edges_merged = nedges;
        ns_lab = ns;
        __syncthreads();
        for (int s = tid; s < ns; s += RS_T)
            if (vdirty[V.label[s]]) { W.sl_moved[s] = 0; ((int *)&W.sl_meta[s])[2] = rounds + gen_off; atomicAnd(&W.sl_hits[s], 0); }
        __syncthreads();
        // ---- members of clusters with 3+ particles are collected for the generic path ----------------------------------------
        unsigned long long *keys = lds_keys;
        for (int s = tid; s < ns; s += RS_T) {
            if (first && s >= wide_ns && V.label[s] == s) atomicAdd(&sh.nclusters, 1);     // (clusters made here; the wide kernel counted its own)
            if (V.size[V.label[s]] >= 3 && vdirty[V.label[s]]) {
                const int k = atomicAdd(&sh.ncomplex, 1);
                const unsigned long long key = ((unsigned long long)(unsigned)V.label[s] << 32) | (unsigned)W.sl_meta[s].x;
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
                const int4 cs = W.cand_s[k];
                if (first && wide && cs.z) continue;                                // emulated (and valid) by the wide kernel
                const int si = cs.x, sj = cs.y;
                if (si < 0 || sj < 0 || si >= ns || sj >= ns) continue;
                if (V.size[V.label[si]] != 2 || !vdirty[V.label[si]]) continue;
                { const int4 c4 = W.cand4[k]; rs_emulate_pair<GEOM>(A, &sh, rs_load_particle(A.S, c4.y), rs_load_particle(A.S, c4.x), c4.y, c4.x, sj, si); }
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
        // ---- validate: every new position of this round against everything outside its cluster -------------------------------
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
                if (rs_hist_gen(hr) == 0) continue;
                const int sme = rs_hist_slot(hr);
                if (idx == W.sl_meta[sme].x) continue;
                const double ex = A.S.x[idx] - hr.x, ey = A.S.y[idx] - hr.y, ez = A.S.z[idx] - hr.z;
                if (ex * ex + ey * ey + ez * ez < cr2i) {
                    const int so = W.slot_of[idx];
                    if (so >= 0 && so < ns && V.label[so] == V.label[sme]) continue;
                    if (so < 0) rs_claim_slot(W, &sh, V.cap, idx);
                    rs_add_edge(W, &sh, W.sl_meta[sme].x, idx);
                }
            }
            for (long long w = tid; w < (long long)(nh - hb) * nh; w += RS_T) {
                const int h = hb + (int)(w / nh), h2 = (int)(w % nh);
                if (h2 >= h) continue;
                const double4 ha = W.hist[h], hb2 = W.hist[h2];
                if (rs_hist_gen(ha) == 0) continue;
                const int s1 = rs_hist_slot(ha), s2 = rs_hist_slot(hb2);
                if (rs_hist_gen(hb2) != W.sl_meta[s2].z) continue;
                if (V.label[s1] == V.label[s2]) continue;
                const double ex = hb2.x - ha.x, ey = hb2.y - ha.y, ez = hb2.z - ha.z;
                if (ex * ex + ey * ey + ez * ez < cr2i) rs_add_edge(W, &sh, W.sl_meta[s1].x, W.sl_meta[s2].x);
            }
        }
        __syncthreads();
        RS_STAMP(4);
        // (the wide kernel's merge edges count as findings of the first round)
        if (!(sh.dirty || (first && wide_dirty)) || sh.ovf || rounds >= RS_MAX_ROUNDS) break;
        __syncthreads();
    }
    if (resolved) rounds = 1;

    const bool ok = !sh.ovf && !(sh.dirty && rounds >= RS_MAX_ROUNDS);
    if (!mono) {
        // the wide commit kernel finishes the sweep
        __syncthreads();
        if (tid == 0) { sh.rounds = rounds; sh.ok = ok; sh.dirty = 0; *ctl = sh; }
        if (A.dbg && tid == 0) { A.dbg[8] += rounds; A.dbg[9] += ncand; A.dbg[10] += sh.ncomplex; A.dbg[11] += 1; }
        return;
    }
    // ---- commit (small sweeps, no-grid mode) ----------------------------------------------------------------------------
    const int ns = sh.nslots < V.cap ? sh.nslots : V.cap;
    const int nh_all = sh.nhist < W.max_hist ? sh.nhist : W.max_hist;
    if (!A.allpairs)
        for (int h = tid; h < nh_all; h += RS_T) W.ov_head[rs_hist_cell(A, A.G, h)] = -1;
    const bool defer = ok && A.defer_commit;        // the next streaming pass reads the slot arrays through slot_of[]
    for (int s = tid; s < ns && !defer; s += RS_T) {
        const int p = W.sl_meta[s].x;
        if (p < 0) continue;                        // a candidate's slot that no particle took
        if (ok && W.sl_moved[s]) {
            rs_apply_slot(W, A.S, s, p);
        }
        W.slot_of[p] = -1;
    }
    if (tid == 0) { sh.nhits = 0; sh.nfp = 0; }
    __syncthreads();
    if (ok) {
        for (int e = tid; e < nh_all; e += RS_T) {
            const int g = W.ev_gen[e];
            if (g == 0) continue;                                           // no completed path at this entry
            const rs_event ev = W.ev[e];
            if (g != W.sl_meta[ev.slot].z) continue;                      // event of an emulation that was redone
            const int owner = ev.which ? ev.i : ev.j;
            if (owner < A.lo || owner >= A.hi) continue;
            amc_emit(A.O, ev.phase, ev.cell, ev.i, ev.j, ev.which, ev.val[0], ev.val[1], ev.val[2], ev.val[3]);
        }
        for (int s2 = tid; s2 < ns; s2 += RS_T) {
            // the counts were updated by atomics (performed in L2): read them there too, not from this CU's L1
            const int hs = __hip_atomic_load(&W.sl_hits[s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (hs & 0xffff) atomicAdd(&sh.nhits, hs & 0xffff);
            if (hs >> 16) atomicAdd(&sh.nfp, hs >> 16);
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

// ---- wide commit (large sweeps) ---------------------------------------------------------------------------------------------
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
            const int p = W.sl_meta[s].x;
            if (p < 0 || W.slot_of[p] != s) continue;   // no particle in the slot / consumed by a streaming pass already
            if (W.sl_moved[s]) {
                rs_apply_slot(W, A.S, s, p);
            }
            W.slot_of[p] = -1;
        }
        return;
    }
    amc_commit_args C;
    C.ctl = (amc_resolve_ctl *)W.ctl; C.hist = W.hist; C.ov_head = W.ov_head; C.ev_gen = W.ev_gen; C.ev = W.ev;
    C.sl_meta = W.sl_meta; C.sl_hits = W.sl_hits; C.sl_moved = W.sl_moved; C.sl_state = W.sl_state; C.slot_of = W.slot_of;
    C.max_slots = W.max_slots; C.max_hist = W.max_hist; C.lo = A.lo; C.hi = A.hi; C.count_pp = A.count_pp;
    C.defer = A.defer_commit; C.nogrid = A.allpairs; C.enabled = 1;
    amc_commit_part(C, A.O, A.G, A.S, gtid, gstride);
}

template <int GEOM>
static void rs_launch_all(amc_ctx *c, const rs_args &A)
{
    if (c->allpairs) {
        AMC_LAUNCH(c, (k_resolve<GEOM, 2>), dim3(1), dim3(RS_T), A);
        return;
    }
    // grid mode: every small component wide first (emulation + its own validation), then the ordered workgroup for what
    // is entangled beyond that.  Launch plan from the candidate count of the most recent sweep the host has seen (a word
    // the kernel writes into host-mapped memory; no synchronisation, it may lag by a step): a small sweep is committed by
    // the workgroup itself, a large one by the wide commit kernel.  Either plan is correct for any count.
    rs_args Aw = A;
    Aw.wide_plan = 1;
    Aw.force_mono = 0;
    {
        // candidates per wave of the wide kernel from the lagging count (with head room; more candidates = more passes)
        const long long lag = c->h_host_ncand ? *c->h_host_ncand : 0;
        const long long nb = amc_clusters_wide_blocks(c);
        long long per = (lag + lag / 4 + nb - 1) / nb;
        Aw.wide_per = (int)std::min<long long>(std::max<long long>(per, 1), 64);
    }
    amc_prof_cancel(c);                 // (the caller's bracket is re-opened below, around the kernel it is named after)
    amc_prof_begin(c, AMC_K_CLUSTERS_WIDE);
    amc_launch_clusters_wide(c, Aw);
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_RESOLVE);
    AMC_LAUNCH(c, (k_resolve<GEOM, 0>), dim3(1), dim3(RS_T), Aw);
    // the commit: deferred results -> it waits for the next streaming pass (or amc_flush); else a kernel of its own, now
    c->commit_defer = A.defer_commit != 0;
    if (A.defer_commit) { c->commit_pending = true; return; }
    amc_prof_end(c);
    amc_prof_begin(c, AMC_K_COMMIT);
    AMC_LAUNCH(c, k_commit, dim3(AMC_COMMIT_BLOCKS), dim3(256), Aw);
}

static rs_args rs_make_args(amc_ctx *c)
{
    rs_args A;
    A.P.collision_range = c->P.collision_range; A.P.argon_mass = c->P.argon_mass; A.P.dx = c->P.dx; A.P.dy = c->P.dy; A.P.dz = c->P.dz;
    A.P.overlap_x = c->P.overlap_x; A.P.overlap_y = c->P.overlap_y; A.P.overlap_z = c->P.overlap_z;
    A.P.nx = c->P.nx; A.P.ny = c->P.ny; A.P.nz = c->P.nz; A.P.geometry = c->P.geometry;
    A.S = c->S; A.G = c->G; A.B = c->B; A.W = c->W; A.O = c->out; A.n = c->n; A.allpairs = c->allpairs ? 1 : 0;
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

hipError_t amc_launch_commit(amc_ctx *c)
{
    rs_args A = rs_make_args(c);
    A.defer_commit = c->commit_defer ? 1 : 0;
    amc_prof_begin(c, AMC_K_COMMIT);
    AMC_LAUNCH(c, k_commit, dim3(AMC_COMMIT_BLOCKS), dim3(256), A);
    amc_prof_end(c);
    return hipGetLastError();
}

hipError_t amc_launch_apply(amc_ctx *c)
{
    rs_args A = rs_make_args(c);
    A.apply_only = 1;
    AMC_LAUNCH(c, k_commit, dim3(8), dim3(256), A);
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
