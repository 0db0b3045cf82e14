// amc_commit_dev.h — the commit of a resolved p-p sweep: completed paths -> histograms / records, collision counters,
// overlay lists emptied, and (unless the results stay in the slot arrays for the next streaming pass) scratch state ->
// particle arrays.  It is wide, order-free work on a few hundred to a few ten thousand entries, and nothing between a
// sweep and the next streaming pass depends on it — so inside a run it RIDES ALONG with that pass (the first blocks of
// k_stream do it, amc_stream.hip) instead of costing a launch of its own; k_commit (amc_resolve.hip) is the stand-alone
// form for a sweep that is followed by something else (a read-back, a stage call).
#pragma once
#include "amc_grid_dev.h"

#define AMC_COMMIT_BLOCKS 64        // blocks of 256 threads that share the commit

struct amc_commit_args {
    amc_resolve_ctl *ctl;           // what the ordered workgroup left: counts, ok flag
    const double4 *hist;
    int *ov_head;
    const int *ev_gen;
    const rs_event *ev;
    const int4 *sl_meta;
    const int *sl_hits;
    const uint8_t *sl_moved;
    const double *sl_state;
    int *slot_of;
    int max_slots, max_hist;
    long long lo, hi;               // owned particle range: a completed path is emitted by the owner of its particle
    int count_pp;                   // this rank adds the sweep's collision count (multi-GPU: the rank that owns particle 0)
    int defer;                      // results stay in the slot arrays (the streaming pass picks them up through slot_of[])
    int nogrid;
    int enabled;
};

AMC_DEV void amc_commit_part(const amc_commit_args &C, const amc_out &O, const amc_grid &G, const amc_state &S, int gtid,
                             int gstride)
{
    amc_resolve_ctl *ctl = C.ctl;
    if (!ctl->active) return;
    const bool ok = ctl->ok && !ctl->ovf;
    const bool defer = ok && C.defer;
    const int ns = ctl->nslots < C.max_slots ? ctl->nslots : C.max_slots;
    const int nh = ctl->nhist < C.max_hist ? ctl->nhist : C.max_hist;
    int my_hits = 0, my_fp = 0;
    for (int s = gtid; s < ns; s += gstride) {
        const int p = C.sl_meta[s].x;
        if (ok && C.count_pp) { const int hs = C.sl_hits[s]; my_hits += hs & 0xffff; my_fp += hs >> 16; }
        if (defer || p < 0) continue;               // (a candidate's slot that no particle took has p = -1)
        if (ok && C.sl_moved[s]) {
            const double *t = C.sl_state + (size_t)s * RS_SLOT_DOUBLES;
            S.x[p] = t[0]; S.y[p] = t[1]; S.z[p] = t[2]; S.vx[p] = t[3]; S.vy[p] = t[4]; S.vz[p] = t[5];
            S.d[p] = t[6]; S.dx[p] = t[7]; S.dy[p] = t[8]; S.dz[p] = t[9]; S.flag[p] = t[10] != 0.0 ? 1 : 0;
        }
        C.slot_of[p] = -1;
    }
    // one atomic per wave instead of one per slot on a single counter word
    for (int o = 32; o > 0; o >>= 1) { my_hits += __shfl_down(my_hits, o, 64); my_fp += __shfl_down(my_fp, o, 64); }
    if ((threadIdx.x & 63) == 0 && my_hits) atomicAdd(&O.banks[amc_bank_id()].n_pp, (unsigned long long)my_hits);
    if ((threadIdx.x & 63) == 0 && my_fp) atomicAdd(&O.banks[amc_bank_id()].n_fp_errors, (unsigned long long)my_fp);
    if (ok)
        for (int e = gtid; e < nh; e += gstride) {
            const int g = C.ev_gen[e];
            if (g == 0) continue;                                       // no completed path at this entry
            const rs_event ev = C.ev[e];
            if (g != C.sl_meta[ev.slot].z) continue;                    // event of an emulation that was redone since
            const int owner = ev.which ? ev.i : ev.j;                   // the particle whose free path completed
            if (owner < C.lo || owner >= C.hi) continue;
            amc_emit(O, ev.phase, ev.cell, ev.i, ev.j, ev.which, ev.val[0], ev.val[1], ev.val[2], ev.val[3]);
        }
    if (!C.nogrid)
        for (int h = gtid; h < nh; h += gstride) {                      // overlay entries of this sweep
            const double4 r = C.hist[h];
            int cx, cy, cz;
            amc_grid_coords(G, r.x, r.y, r.z, cx, cy, cz);
            C.ov_head[amc_grid_cell(G, cx, cy, cz, nullptr)] = -1;
        }
    if (gtid == 0) {
        amc_dev_counters *cnt = O.cnt;
        cnt->n_candidates += (unsigned long long)ctl->ncand;
        cnt->n_clusters += (unsigned long long)ctl->nclusters;
        cnt->n_rounds += (unsigned long long)ctl->rounds;
        if (!ok) cnt->flags |= 4ULL;
        ctl->lazy_ns = defer ? ns : 0;
    }
}
