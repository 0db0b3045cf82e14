// amc_clusters.hip — the wide half of the ordered p-p resolve: every SMALL connected component of the candidate graph is
// emulated here, spread over the whole chip, and validates its own new positions; the single ordered workgroup
// (k_resolve, amc_resolve.hip) is left with what is entangled beyond that — components larger than CW_MAXM particles,
// and clusters whose validation found an outsider (merge, re-emulate, re-validate) — plus the commit of a small sweep.
//
// Reference semantics kept (Pore:168-241 in-place i>j order, Pore:522-530 group / cell order, Cube:231-238): a cluster
// is emulated literally, restricted to its members, exactly as the ordered workgroup would do it (rs_emulate_pair /
// rs_emulate_coop / rs_emulate_generic are shared); what is added here is only WHO does it and WHEN it is validated.
//
//   1. one lane per candidate k.  It reads the candidate record (i, j, the candidates before it in the lists of i and j)
//      and the candidate-indexed mark the detect kernel sets when a LATER candidate shares a particle with it
//      (amc_push_candidate, amc_grid.hip).  Nobody before, nobody after  =>  isolated pair: the lane owns it and never
//      touches the particle-indexed graph heads.  Nobody before and ONE candidate after (the mark names it): one more
//      record and its mark tell whether the component is that chain of two — three particles, known after one round
//      trip.  Otherwise the lane walks the component from the heads; it gives up as
//      soon as it meets a candidate with a lower index (the lowest candidate's lane owns the component) or when the
//      component exceeds CW_MAXM particles / CW_MAXC candidates (left to the ordered workgroup: nobody marks its
//      candidates done).
//   2. no allocation: candidate k brings slots 2k, 2k + 1 and the history / event pairs 4k, 4k + 2 (zeroed by the detect
//      kernel when it pushed k); a cluster uses what its candidates brought.  Only a pulled-in particle's slot and a hit
//      beyond 2 x candidates take entries from counters (behind 2 ncand / 4 ncand); such a cluster cannot be published
//      here and is flagged for the ordered workgroup, which redoes it — rare.
//   3. isolated pairs and 3-particle clusters are emulated by their lanes in registers; larger clusters one after the
//      other by the whole wave (working set in LDS, members in ascending particle index).
//   4. validation in two halves.  The new positions stay in LDS first and are probed against the pre-sweep positions of
//      everything outside the cluster (the detection grid's lists).  A hit on a particle in no candidate pulls it in
//      (compare-and-swap on slot_of) and the grown cluster is emulated again from the untouched pre-sweep state, up to
//      CW_ITERS emulations; nothing of a superseded emulation was ever visible to another wave.  (A pair with one hit that
//      pulls ONE particle in is not started over: the grown cluster's emulation reaches that hit with the same operands, so
//      it continues from the hit's result under the same round tag, rs_first_hit.)  Then the FINAL
//      emulation is published — write-through history records, each pushed on the overlay list of its grid cell by a
//      compare-and-swap on the list head (a reader never meets a half-linked entry) — and only after ALL pushes of the
//      wave have returned are the positions probed against the other clusters' new positions (the overlay lists, read
//      at agent scope).  Two clusters whose new positions conflict are both pushed before either probes, or one probes
//      after the other's push: at least one of them sees the other.  That hit becomes a merge edge for the ordered
//      workgroup, which merges, re-emulates from the pre-sweep state and validates again.
//
// Bound: latency — ~6 dependent memory round trips of 0.5-0.6 us and ~2,000 instructions of one lane per candidate
//  — which is why the candidates are spread as thinly as the launch allows: CW_BLOCKS one-wave
// workgroups, ceil(ncand / CW_BLOCKS) candidates per wave.
#include <stdlib.h>

#define RS_WAVE_SYNC_LDS_ONLY 1     // one wave per cluster, working set in LDS (see rs_wave_sync)
#include "amc_resolve_dev.h"

#define CW_MAXM 16          // particles of a component handled here
#define CW_MAXC 24          // its candidates
#define CW_ITEMS 192        // new positions one wave can publish + probe per pass (two per hit)
#define CW_BLOCKS 512       // waves of the launch
#define CW_WPB 1            // waves per workgroup; each wave works on its own (its own part of the LDS, wave-level synchronisation
                            // only).  Measured with 4 (128 workgroups of four waves, sharing instruction cache and LDS of a CU):
                            // cube N = 1e5 51.0 instead of 48.0 us per step, pore N = 1e6 139.7 instead of 136 — one wave per CU it is

AMC_DEV int cw_adj_head(const amc_resolve_ws &W, unsigned int epoch, int p)
{
    const unsigned long long v = W.adj_head[p];
    return ((unsigned int)(v >> 32) == epoch) ? (int)(unsigned int)(v & 0xffffffffULL) : -1;
}

#define CW_ITERS 3          // emulations of one cluster in this kernel: the first + two after it pulled particles in
#define CW_PULLS 4          // particles one cluster can pull in per validation

AMC_DEV void cw_init_slot(const amc_resolve_ws &W, int s, int p, int label, int gen, bool fresh = false)
{
    W.slot_of[p] = s; W.sl_moved[s] = 0;
    // (particle, label, round, -) written through: a prober of another workgroup that meets one of this slot's history
    // entries compares rounds
    unsigned long long *m = (unsigned long long *)&W.sl_meta[s];
    __hip_atomic_store(m + 0, ((unsigned long long)(unsigned int)label << 32) | (unsigned int)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(m + 1, (unsigned long long)(unsigned int)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!fresh) atomicAnd(&W.sl_hits[s], 0);    // (an atomic, like the increments that follow; a candidate's own slot was
                                                // zeroed by the detect kernel when it pushed the candidate)
}

AMC_DEV double4 cw_load_hist(const amc_resolve_ws &W, int h)
{
    const double *d = (const double *)&W.hist[h];
    double4 r;
    r.x = __hip_atomic_load(d + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.y = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.z = __hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.w = __hip_atomic_load(d + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
}

struct cw_lds {
    int mem[64][CW_MAXM];       // per owner lane: the particles of its cluster (ascending when emulated)
    int msl[64][CW_MAXM];       // and their slots
    int cnd[64][CW_MAXC];       // its candidates: candidate c brings slots 2c, 2c + 1 and the history pairs 4c, 4c + 2
    int pull[64][CW_PULLS], psl[64][CW_PULLS];     // particles (and their new slots) the last validation pulled in
    int nm[64], nc[64], lab[64], npull[64];         // members, candidates, cluster label (= first slot), pulls
    int it0[64];                // first pair of work items of the owner's current emulation
    int used[64];               // history pairs the running emulation has taken
    int redo[64];               // the cluster must be emulated (again) by the wave
    int gen[64];                // emulations done
    cw_item item[CW_ITEMS];
    int next[CW_ITEMS];         // overlay `next` of every published item (own entries are stepped over without a load)
    int nitems;
    int unv[64];                // the running emulation of the owner got entries the wave cannot publish
    double pool_d[10][CW_MAXM];
    int pool_tmp[CW_MAXM], pool_pidx[CW_MAXM], pool_slot[CW_MAXM];
    uint8_t pool_flag[CW_MAXM], pool_moved[CW_MAXM];
};

// The validation of a new position has two halves.  GRID half: the pre-sweep positions of everything outside the cluster
// (the detection grid's lists) — needs nothing from other workgroups, so it runs right after the emulation, before
// anything is published: a particle that is in no candidate and that nobody else has taken JOINS the cluster (the owner
// emulates again with it, and what it had emulated so far is simply dropped: it was never visible to anybody); anything
// else becomes a merge edge for the ordered workgroup.  OVERLAY half: the other clusters' new positions, after the
// cluster's final positions have been published (publish-then-probe, see the header).  One call handles ONE cell of the
// position's box; the (up to eight) cells go to different lanes while the wave has lanes to spare — the probe is a chain
// of dependent round trips plus a few hundred instructions per cell, and a lone wave issues one instruction every ~2 ns.
AMC_DEV void cw_probe_grid(const rs_args &A, rs_shared *wc, cw_lds &L, const cw_item &me, int cell, int s_off)
{
    const amc_resolve_ws &W = A.W;
    const int own = me.own, nm = L.nm[own], lab = L.lab[own];
    const double x = me.x, y = me.y, z = me.z;
    int q = amc_list_head(A.B, cell);
    while (q >= 0) {
        const amc_rec r = A.B.rec[q];
        const int idx = amc_node_particle(A.B, q);
        q = amc_rec_next(r);
        double rx, ry, rz;
        amc_rec_pos(A.G, r, rx, ry, rz);
        const double ax = rx - x, ay = ry - y, az = rz - z;
        if (!(ax * ax + ay * ay + az * az < A.G.cr2_probe)) continue;      // (single-precision record: widened test)
        bool mine = false;
        for (int m = 0; m < nm; m++) mine |= L.mem[own][m] == idx;
        if (mine) continue;
        if (cw_adj_head(W, A.sweep_epoch, idx) < 0) {
            // in no candidate: pull it into this cluster, unless somebody else got it first
            const int tag = -(lab + 2);
            const int old = atomicCAS(&W.slot_of[idx], -1, tag);
            if (old == tag) continue;                           // another position of my cluster found it too
            if (old == -1) {
                const int s = s_off + atomicAdd(&wc->nslots, 1);
                const int k = atomicAdd(&L.npull[own], 1);
                W.victim[idx] = A.sweep_epoch;                  // (in no candidate: an overlapped streaming pass has advanced it)
                if (s < W.max_slots && k < CW_PULLS) {
                    L.pull[own][k] = idx; L.psl[own][k] = s;   // (slot_of keeps the tag until the owner initialises the slot)
                    L.redo[own] = 1;
                    continue;
                }
                if (s >= W.max_slots) wc->ovf = 1;
                else { W.sl_meta[s] = make_int4(idx, s, 0, 0); W.sl_moved[s] = 0; atomicAnd(&W.sl_hits[s], 0); }
                W.slot_of[idx] = s < W.max_slots ? s : -1;      // too many at once: a plain merge edge instead
            }
        }
        rs_add_edge(W, wc, me.p, idx);
    }
}

AMC_DEV void cw_probe_overlay(const rs_args &A, rs_shared *wc, cw_lds &L, const cw_item &me, int cell, double cr2i, int h_off)
{
    const amc_resolve_ws &W = A.W;
    const int own = me.own, nm = L.nm[own];
    const double x = me.x, y = me.y, z = me.z;
    // the published entries of my own cluster — the pairs its candidates brought along, in the order its final emulation
    // took them — are stepped over through the `next` values their pushes returned (LDS), without a load
    auto skip_own = [&](int h2) {
        for (int steps = 0; h2 >= 0 && h2 < h_off && steps < CW_ITEMS; steps++) {
            const int c = h2 >> 2, second = (h2 >> 1) & 1, ncs = L.nc[own];
            int t = -1;
            for (int q = 0; q < ncs; q++)
                if (L.cnd[own][q] == c) { t = L.it0[own] + 2 * (q + second * ncs) + (h2 & 1); break; }
            // Hit number q + second * ncs of my final emulation sits that many pairs behind my first one ONLY if no other
            // lane appended in between — the lanes of a wave emulate in lockstep and share the list, so the pairs of two
            // multi-hit clusters interleave.  The item says whose entry it holds: anything else is left to the walk below,
            // which treats it like anybody's entry (a load more, the same result).
            if (t < 0 || t >= CW_ITEMS || L.item[t].h != h2) break;
            h2 = L.next[t];
        }
        return h2;
    };
    int h2 = skip_own(__hip_atomic_load(&W.ov_head[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    int guard = 0;
    while (h2 >= 0) {
        if (++guard > W.max_hist) { wc->ovf = 1; break; }      // (a list cannot be longer than there are entries: reported, not spun on)
        const double4 o = cw_load_hist(W, h2);
        const int nx = __hip_atomic_load(&W.ov_next[h2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h2 = skip_own(nx);
        const double ax = o.x - x, ay = o.y - y, az = o.z - z;
        if (!(ax * ax + ay * ay + az * az < cr2i)) continue;
        const int s2 = rs_hist_slot(o);
        bool mine = false;
        for (int m = 0; m < nm; m++) mine |= L.msl[own][m] == s2;
        if (mine) continue;
        // (only final emulations are ever published, but the ordered workgroup's rounds are told apart the same way)
        if (rs_hist_gen(o) != __hip_atomic_load(((int *)&W.sl_meta[s2]) + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
        rs_add_edge(W, wc, me.p, -(s2 + 2));                    // (the other end as a slot)
    }
}

// Before an owner's cluster is emulated (again): the particles its last validation pulled in become members (kept in
// ascending particle index, slots alongside).  The emulation starts over with the history pairs its candidates brought
// along: what the dropped emulation wrote there was never published.  Run by ONE lane per owner.  Returns the number of members.
AMC_DEV int cw_prepare(const amc_resolve_ws &W, rs_shared *wc, cw_lds &L, int own, int h_off)
{
    int m = L.nm[own];
    const int lab = L.lab[own];
    const int np = L.npull[own] < CW_PULLS ? L.npull[own] : CW_PULLS;
    L.unv[own] = 0;
    for (int e = 0; e < np; e++) {
        const int v = L.pull[own][e], sv = L.psl[own][e];
        if (m == CW_MAXM) {             // no room: it keeps its slot (same label) and the ordered workgroup takes over
            cw_init_slot(W, sv, v, lab, 0);
            L.unv[own] = 2;
            continue;
        }
        int b = m - 1;
        while (b >= 0 && L.mem[own][b] > v) { L.mem[own][b + 1] = L.mem[own][b]; L.msl[own][b + 1] = L.msl[own][b]; b--; }
        L.mem[own][b + 1] = v; L.msl[own][b + 1] = sv;
        m++;
    }
    L.nm[own] = m; L.npull[own] = 0;
    L.used[own] = 0;
    L.it0[own] = -1;
    return m;
}

AMC_DEV void cw_wide_hooks(rs_wide &wd, cw_lds &L, int own, int h_off)
{
    const int g = L.gen[own];
    wd.cnd = L.cnd[own]; wd.ncnd = L.nc[own];
    wd.used = &L.used[own]; wd.h_off = h_off;
    wd.items = L.item; wd.nitems = &L.nitems; wd.cap = CW_ITEMS;
    wd.own = own; wd.gen = g + 1; wd.it0 = &L.it0[own]; wd.unval = &L.unv[own];
}

// slots of the cluster's candidates (first emulation only)
AMC_DEV void cw_candidate_slots(const amc_resolve_ws &W, cw_lds &L, int own, int e)
{
    const int c = L.cnd[own][e], m = L.nm[own];
    const int4 cc = W.cand4[c];
    int ai = 0, aj = 0;
    for (int t = 0; t < m; t++) { if (L.mem[own][t] == cc.x) ai = t; if (L.mem[own][t] == cc.y) aj = t; }
    W.cand_s[c] = make_int4(L.msl[own][ai], L.msl[own][aj], 1, 0);
}

template <int GEOM, bool DBG>
__global__ __launch_bounds__(64 * CW_WPB) void k_clusters_wide(rs_args A_in_kernarg)
{
    RS_STAGE_ARGS(A);
    const amc_resolve_ws &W = A.W;
    rs_shared *wc = (rs_shared *)W.wctl;
    __shared__ cw_lds L_all[CW_WPB];
    cw_lds &L = L_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * CW_WPB;
    const int wave_id = blockIdx.x * CW_WPB + (threadIdx.x >> 6);
    long long *const dbg__ = DBG ? A.dbg : nullptr;     // (the phase timers are compiled out of the product's instantiation)
    const long long t_enter__ = dbg__ ? wall_clock64() : 0;
    const int per = A.wide_per;         // candidates per wave and pass, fixed by the host: the first pass's candidate is
                                        // known before the sweep's candidate count has arrived
    // speculative: record and state of my first candidate (valid memory for any k below the capacity)
    const int k_first = wave_id * per + lane;
    int4 c4_first = make_int4(0, 0, -1, -1);
    unsigned long long mark_first = 0;
    if (lane < per && k_first < W.max_cand) { c4_first = W.cand4[k_first]; mark_first = W.cand_mark[k_first]; }
    int ncand = (int)A.O.cnt->cand_count;
    if (ncand > W.max_cand) ncand = W.max_cand;
    if (wave_id == 0 && lane == 0) { wc->active = 1; wc->ncand = ncand; }
    if (ncand == 0) return;
    const int s_off = 2 * ncand;        // first counter-allocated slot of this sweep (candidate k owns slots 2k, 2k + 1)
    const int h_off = 4 * ncand;        // first counter-allocated history entry (candidate k owns the pairs 4k and 4k + 2)
    const double cr2i = A.P.collision_range * A.P.collision_range * AMC_CR2_INFLATE;
    rs_work K;
    K.x = L.pool_d[0]; K.y = L.pool_d[1]; K.z = L.pool_d[2]; K.vx = L.pool_d[3]; K.vy = L.pool_d[4]; K.vz = L.pool_d[5];
    K.d = L.pool_d[6]; K.dx = L.pool_d[7]; K.dy = L.pool_d[8]; K.dz = L.pool_d[9];
    K.tmp = L.pool_tmp; K.pidx = L.pool_pidx; K.slot = L.pool_slot; K.flag = L.pool_flag; K.moved = L.pool_moved;

    // phase timers (diagnostic, AMC_DEBUG_RESOLVE=1): kept in registers and added to the debug buffer when the wave ends —
    // an atomic per stamp would sit in the wave's memory queue in front of the loads it is supposed to time
    long long t_last = t_enter__;
    long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CW_STAMP(slot)                                                                     \
    do {                                                                                   \
        if (timed__) {                                                                     \
            const long long now__ = wall_clock64();                                        \
            t_acc[slot] += now__ - t_last;                                                 \
            t_last = now__;                                                                \
        }                                                                                  \
    } while (0)
    const bool timed__ = DBG && dbg__ && lane == 0 && wave_id * per < ncand;     // waves with work in their first pass
    if (DBG && dbg__) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CW_STAMP(7); }      // entry -> candidate count and my first candidate have arrived
    int cat__ = 0;
    long long d_cont = 0, d_three = 0, d_t_load = 0, d_t_emu = 0;      // (diagnostic: the three-particle path of lane 0)
    for (int k0 = wave_id * per; k0 < ncand; k0 += nwaves * per) {       // wave-uniform trip count
        const int k = k0 + lane;
        const bool valid = lane < per && k < ncand;
        if (lane == 0) L.nitems = 0;
        // ---- 1. my candidate and the graph around it -----------------------------------------------------------------
        int4 c4 = make_int4(0, 0, -1, -1);
        int head_i = -1, head_j = -1;
        amc_particle pre_j, pre_i;          // state of both particles
        bool fh_ok = false;                 // a pair's only hit can be taken over when its validation pulls a third particle in
        int fh_it0 = -1, fh_pj = -1, fh_pi = -1, fh_sj = -1, fh_si = -1;
        bool iso = false;
        // a chain of two candidates (k, then k2 on one of my particles, nothing else anywhere): the second candidate
        // comes with my mark, so the component is known after ONE round trip (its record and its mark)
        int k2 = -1;
        int4 c4b = make_int4(0, 0, -1, -1);
        unsigned long long mkb = 0;
        if (valid) {
            c4 = (k == k_first) ? c4_first : W.cand4[k];
            const unsigned long long mk = (k == k_first) ? mark_first : W.cand_mark[k];
            const bool marked = (unsigned int)(mk >> 32) == A.sweep_epoch;
            // alone on both particles: nobody before it in either list, nobody displaced it from either head (the detect
            // kernel marks the displaced candidate) — decided from candidate-indexed words only
            iso = c4.z < 0 && c4.w < 0 && !marked;
            if (c4.z < 0 && c4.w < 0 && marked) {
                const unsigned int succ = (unsigned int)(mk & 0xffffffffULL);
                if (succ != AMC_MARK_MULTI && (int)succ > k && (int)succ < ncand) k2 = (int)succ;
            }
            if (k2 >= 0) { c4b = W.cand4[k2]; mkb = W.cand_mark[k2]; }
            else if (!iso) {                // part of a larger component: the walk below starts at the particles' heads
                head_i = cw_adj_head(W, A.sweep_epoch, c4.x);
                head_j = cw_adj_head(W, A.sweep_epoch, c4.y);
            }
        }
        CW_STAMP(9);
        if (valid) {
            pre_j = rs_load_particle(A.S, c4.y);
            pre_i = rs_load_particle(A.S, c4.x);
        }
        if (DBG && dbg__) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CW_STAMP(8); }      // -> both particles (and the heads) have arrived
        CW_STAMP(0);
        bool owner = valid && !iso;
        int nm = 2, nc = 1;
        int *mem = L.mem[lane], *cnd = L.cnd[lane];
        mem[0] = c4.y; mem[1] = c4.x; cnd[0] = k;
        bool chain2 = false;
        if (k2 >= 0) {
            // k2 is the head of both its particles' lists (nobody displaced it), my candidate is the one before it on the
            // shared particle and there is nobody before it on the other: the component is {k, k2}, three particles
            const bool via_x = c4b.z == k && c4b.w < 0, via_y = c4b.w == k && c4b.z < 0;
            chain2 = (unsigned int)(mkb >> 32) != A.sweep_epoch && (via_x || via_y);
            if (chain2) { mem[2] = via_x ? c4b.y : c4b.x; cnd[1] = k2; nm = 3; nc = 2; }
            else {                          // something else: the general walk, from the heads
                head_i = cw_adj_head(W, A.sweep_epoch, c4.x);
                head_j = cw_adj_head(W, A.sweep_epoch, c4.y);
            }
        }
        if (owner && !chain2) {
            for (int cur = 0; cur < nm && owner; cur++) {
                const int p = mem[cur];
                int c = cur == 0 ? head_j : (cur == 1 ? head_i : cw_adj_head(W, A.sweep_epoch, p));
                while (c >= 0) {
                    if (c < k) { owner = false; break; }            // the component belongs to a lower candidate's lane
                    const int4 r = (c == k) ? c4 : W.cand4[c];
                    const int q = (r.x == p) ? r.y : r.x;
                    const int nx = (r.x == p) ? r.z : r.w;
                    bool seen = false;
                    for (int e = 0; e < nc; e++) seen |= cnd[e] == c;
                    if (!seen) {
                        if (nc == CW_MAXC) { owner = false; wc->changed = 1; break; }       // too large for this kernel: left to the ordered workgroup
                        cnd[nc++] = c;
                    }
                    seen = false;
                    for (int e = 0; e < nm; e++) seen |= mem[e] == q;
                    if (!seen) {
                        if (nm == CW_MAXM) { owner = false; wc->changed = 1; break; }
                        mem[nm++] = q;
                    }
                    c = nx;
                }
            }
        }
        if (owner)                                                  // members in ascending particle index (Pore:538)
            for (int a = 1; a < nm; a++) {
                const int v = mem[a];
                int b = a - 1;
                while (b >= 0 && mem[b] > v) { mem[b + 1] = mem[b]; b--; }
                mem[b + 1] = v;
            }
        CW_STAMP(1);
        // ---- 2. no allocation: member t of a cluster takes slot 2 c + (t & 1) of the cluster's candidate c = cnd[t / 2]
        // (a cluster of nc candidates has at most nc + 1 particles), the q-th hit the history pair of candidate cnd[q] ----
        const bool take = iso || owner;
        L.nm[lane] = nm; L.nc[lane] = nc; L.lab[lane] = 2 * k; L.npull[lane] = 0;
        for (int m = 0; m < nm; m++) L.msl[lane][m] = 2 * cnd[m >> 1] + (m & 1);
        L.it0[lane] = -1;
        L.redo[lane] = take ? 1 : 0;
        L.gen[lane] = 0;
        if (timed__) cat__ = !take ? 3 : (nm == 2 ? 0 : (nm == 3 ? 1 : 2));
        {
            // statistics: one counter per bank of waves (2000 waves adding to ONE word are a 24 us chain of same-address atomics)
            const int ncl = __popcll(__ballot(take));
            if (lane == 0 && ncl) atomicAdd(&W.wctl[32 + (wave_id & 15)], ncl);
        }
        rs_wave_sync();
        CW_STAMP(2);
        int first_item = 0;
        for (int iter = 0; iter < CW_ITERS; iter++) {
            // ---- 3a. clusters of two or three particles: one lane each, everything in registers ----------------------------------
            rs_wave_sync();
            int my_m = 0;
            if (L.redo[lane]) my_m = cw_prepare(W, wc, L, lane, h_off);
            if (L.redo[lane] && my_m <= 3) {
                const int g = L.gen[lane], lab = L.lab[lane];
                rs_wide wd;
                cw_wide_hooks(wd, L, lane, h_off);
                if (my_m == 2) {
                    // (an isolated pair; only ever emulated once: a pair that pulls a particle in comes back as three)
                    const int pj = mem[0], pi = mem[1], sj = L.msl[lane][0], si = L.msl[lane][1];
                    cw_init_slot(W, sj, pj, lab, g + 1, true);
                    cw_init_slot(W, si, pi, lab, g + 1, true);
                    W.cand_s[k] = make_int4(si, sj, 1, 0);
                    CW_STAMP(11);
                    amc_particle p1 = pre_j, p2 = pre_i;
                    const bool mv = rs_emulate_pair_io<GEOM>(A, wc, p1, p2, pj, pi, sj, si, &wd);
                    // exactly one hit, and it moved the pair (their state after it is in their slots)
                    fh_ok = mv && L.used[lane] == 1 && !L.unv[lane] && GEOM != AMC_GEOM_CELL;
                    fh_pj = pj; fh_pi = pi; fh_sj = sj; fh_si = si;
                    fh_it0 = L.it0[lane];
                    CW_STAMP(4);
                    L.gen[lane] = g + 1;
                } else {
                    // A pair that pulled ONE particle in continues from its hit (rs_first_hit) under the round tag it has: the
                    // hit's entries, events and work items stay valid (their positions have been probed against the grid
                    // already), only the new member gets a slot.  Anything else starts over under the next tag.
                    const bool cont = fh_ok && g == 1;
                    fh_ok = false;
                    const long long d_t0 = timed__ ? wall_clock64() : 0;
                    amc_particle q[3];
                    int pidx[3], slot[3];
                    bool moved[3] = {false, false, false};
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        pidx[a] = mem[a]; slot[a] = L.msl[lane][a];
                        // (first emulation: the candidate's two particles were loaded with it)
                        if (g == 0 && pidx[a] == c4.y) q[a] = pre_j;
                        else if (g == 0 && pidx[a] == c4.x) q[a] = pre_i;
                        else q[a] = rs_load_particle(A.S, pidx[a]);
                    }
                    rs_first_hit fh;
                    fh.pj = cont ? fh_pj : -1; fh.pi = cont ? fh_pi : -1;
                    if (cont) { fh.p1 = rs_load_slot(W, fh_sj); fh.p2 = rs_load_slot(W, fh_si); }
                    else { fh.p1 = q[0]; fh.p2 = q[0]; }
                    const int tag = cont ? g : g + 1;
#pragma unroll
                    for (int a = 0; a < 3; a++)
                        if (!cont || (pidx[a] != fh_pj && pidx[a] != fh_pi)) cw_init_slot(W, slot[a], pidx[a], lab, tag);
                    if (cont) { wd.gen = tag; L.used[lane] = 1; L.it0[lane] = fh_it0; }
                    if (g == 0)
                        for (int e = 0; e < L.nc[lane]; e++) cw_candidate_slots(W, L, lane, e);
                    long long d_t1 = 0;
                    if (DBG && timed__) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); d_t1 = wall_clock64(); }
                    rs_emulate_small<GEOM, 3>(A, wc, q, pidx, slot, moved, &wd, cont, fh);
                    if (DBG && timed__) { d_three++; d_cont += cont; d_t_load += d_t1 - d_t0; d_t_emu += wall_clock64() - d_t1; }
#pragma unroll
                    for (int a = 0; a < 3; a++)
                        if (moved[a]) rs_store_slot(W, slot[a], q[a]);
                    L.gen[lane] = tag;
                }
                if (L.unv[lane]) rs_add_edge(W, wc, mem[0], mem[0]);     // (self edge: the ordered workgroup redoes this cluster)
                L.redo[lane] = 0;
            }
            // (a cluster whose pulled particle did not fit goes through the cooperative path below all the same: its members,
            // the ones just pulled in too, get their slots there, and the self edge hands it to the ordered workgroup)
            CW_STAMP(3);
            // ---- 3b. larger clusters, one after the other by the whole wave (working set in LDS) ---------------------------------
            rs_wave_sync();
            unsigned long long todo = __ballot(L.redo[lane] != 0);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int g = L.gen[src], ncs = L.nc[src], lab = L.lab[src];
                const int m = L.nm[src];
                if (lane < m) {
                    const int p = L.mem[src][lane], sl = L.msl[src][lane];
                    const amc_particle q = rs_load_particle(A.S, p);
                    rs_store_work(K, lane, q);
                    K.moved[lane] = 0; K.pidx[lane] = p; K.slot[lane] = sl;
                    cw_init_slot(W, sl, p, lab, g + 1);
                }
                if (g == 0)
                    for (int e = lane; e < ncs; e += 64) cw_candidate_slots(W, L, src, e);
                rs_wave_sync();
                rs_wide wd;
                cw_wide_hooks(wd, L, src, h_off);
                if (m <= RS_COOP_MAX) rs_emulate_coop(A, wc, K, 0, m, &wd);
                else if (lane == 0) rs_emulate_generic(A, wc, K, 0, m, &wd);
                rs_wave_sync();
                if (lane < m && K.moved[lane]) rs_store_slot(W, K.slot[lane], rs_load_work(K, lane));
                if (lane == 0) {
                    if (L.unv[src]) rs_add_edge(W, wc, K.pidx[0], K.pidx[0]);  // (self edge: the ordered workgroup redoes it)
                    L.gen[src] = g + 1;
                    L.redo[src] = 0;
                }
                rs_wave_sync();
            }
            CW_STAMP(4);
            // ---- 4a. GRID half of the validation, straight after the emulation (nothing has been published yet) ------------------
            rs_wave_sync();
            const int nit = L.nitems < CW_ITEMS ? L.nitems : CW_ITEMS;
            {
                // the cells of a position's box go to different lanes while the wave has lanes to spare
                const int nnew = nit - first_item;
                const int lpi = nnew <= 8 ? 8 : (nnew <= 16 ? 4 : (nnew <= 32 ? 2 : 1));       // lanes per item
                const int sub = lane % lpi;
                for (int t = first_item + lane / lpi; t < nit; t += 64 / lpi) {
                    const cw_item it = L.item[t];
                    if (it.pad & 1) continue;
                    for (int c = sub; c < 8; c += lpi) {
                        const int cell = amc_grid_box_cell(A.G, it.x, it.y, it.z, A.G.cr_probe, c);
                        if (cell >= 0) cw_probe_grid(A, wc, L, it, cell, s_off);
                    }
                }
            }
            first_item = nit;
            CW_STAMP(6);
            // a cluster that pulled particles in is emulated again from the untouched pre-sweep state (what it emulated so far
            // was never published: it is simply dropped), unless it has had its turns: then the particles still get their
            // slots and the ordered workgroup takes over
            if (!__ballot(L.redo[lane] != 0)) break;
            if (timed__ && L.redo[0]) cat__ |= 4;
            if (iter + 1 == CW_ITERS || L.nitems >= CW_ITEMS - 8) {
                if (L.redo[lane]) {
                    const int np = L.npull[lane] < CW_PULLS ? L.npull[lane] : CW_PULLS;
                    for (int e = 0; e < np; e++) cw_init_slot(W, L.psl[lane][e], L.pull[lane][e], L.lab[lane], 0);
                    rs_add_edge(W, wc, L.mem[lane][0], L.mem[lane][0]);
                    L.redo[lane] = 0;
                }
                break;
            }
        }
        // ---- 4b. publish the positions of every cluster's FINAL emulation, then the OVERLAY half of their validation ------------
        rs_wave_sync();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the history records (write-through stores) have left
        const int nit = L.nitems < CW_ITEMS ? L.nitems : CW_ITEMS;
        for (int t0 = 0; t0 < nit; t0 += 64) {
            const int t = t0 + lane;
            if (t < nit) L.next[t] = -1;
            if (t < nit) {
                const cw_item it = L.item[t];
                if (!(it.pad & 1) && (it.pad >> 1) == L.gen[it.own]) {  // (an item of a dropped emulation is skipped)
                    int cx, cy, cz;
                    amc_grid_coords(A.G, it.x, it.y, it.z, cx, cy, cz);
                    const int cell = amc_grid_cell(A.G, cx, cy, cz, nullptr);
                    int expected = -1;                              // (the record was stored with next = -1)
                    for (;;) {
                        const int old = atomicCAS(&W.ov_head[cell], expected, it.h);
                        if (old == expected) break;
                        expected = old;
                        __hip_atomic_store(&W.ov_next[it.h], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    L.next[t] = expected;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every push of this wave has returned
        rs_wave_sync();
        CW_STAMP(5);
        {
            const int lpi = nit <= 8 ? 8 : (nit <= 16 ? 4 : (nit <= 32 ? 2 : 1));
            const int sub = lane % lpi;
            for (int t = lane / lpi; t < nit; t += 64 / lpi) {
                const cw_item it = L.item[t];
                if ((it.pad & 1) || (it.pad >> 1) != L.gen[it.own]) continue;
                for (int c = sub; c < 8; c += lpi) {
                    const int cell = amc_grid_box_cell(A.G, it.x, it.y, it.z, A.G.cr_probe, c);
                    if (cell >= 0) cw_probe_overlay(A, wc, L, it, cell, cr2i, h_off);
                }
            }
        }
        rs_wave_sync();
        CW_STAMP(10);
    }
    if (timed__) {
        // Every wave keeps its figures in 64 words of its own (plain read-modify-write: a wave id occurs once per launch and
        // launches are serial).  The first version added them to shared words with atomics: a few hundred waves hitting one
        // line jam that memory channel for microseconds — it showed up as a 4 us "stall" of whatever the OTHER waves did next.
        long long *R = dbg__ + 128 + 128 * (long long)wave_id;
        const long long t_exit = wall_clock64(), life = t_exit - t_enter__;
        R[0] = t_enter__; R[1] = t_exit; R[2] = A.sweep_epoch;          // (this launch: the ordered workgroup forms the span)
        R[3] += 1;
        if (life > R[4]) R[4] = life;
        R[5] += d_three; R[6] += d_cont; R[7] += d_t_load; R[124] += d_t_emu;   // (R[124..127]: the tail of the last category's figures, never reached)
        R[8 + cat__] += life; R[16 + cat__] += 1;                       // by kind of wave
        for (int e = 0; e < 12; e++) R[32 + 12 * cat__ + e] += t_acc[e];
        if (cat__ == 0) { int bk = (int)(life / 250); if (bk > 7) bk = 7; R[24 + bk] += 1; }
    }
}

int amc_clusters_wide_blocks(amc_ctx *c)
{
    // AMC_CW_BLOCKS (read when the context is created): fewer waves = more clusters per wave — tests use it to put many
    // clusters, multi-hit ones among them, into the same wave at sizes the oracle handles in seconds
    const int nb_env = c->cw_blocks_env;
    if (nb_env > 0) return (nb_env + CW_WPB - 1) / CW_WPB * CW_WPB;
    return CW_BLOCKS;
}

hipError_t amc_launch_clusters_wide(amc_ctx *c, const rs_args &A)
{
    const int nb = amc_clusters_wide_blocks(c);
    const dim3 grid((nb + CW_WPB - 1) / CW_WPB), block(64 * CW_WPB);
    const bool cube = c->P.geometry == AMC_GEOM_CUBE;
    if (A.dbg) {        // AMC_DEBUG_RESOLVE=1: the instantiation with the phase timers
        if (cube) AMC_LAUNCH(c, (k_clusters_wide<AMC_GEOM_CUBE, true>), grid, block, A);
        else AMC_LAUNCH(c, (k_clusters_wide<AMC_GEOM_PORE, true>), grid, block, A);
    } else {
        if (cube) AMC_LAUNCH(c, (k_clusters_wide<AMC_GEOM_CUBE, false>), grid, block, A);
        else AMC_LAUNCH(c, (k_clusters_wide<AMC_GEOM_PORE, false>), grid, block, A);
    }
    return hipGetLastError();
}
