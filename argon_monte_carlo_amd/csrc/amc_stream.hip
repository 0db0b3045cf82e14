// amc_stream.hip — the per-particle streaming stage: drift + free-path accumulation + wall cases + bounds check,
// fused into ONE pass over the SoA state (reference: Cube:179-226, Pore:426-485 + 354-375/512).
//
// Roofline: HBM-bound.  Algorithmic traffic per particle-step: read pos 24 + vel 24 + 4 accumulators 32 + flag 1
// = 81 B, write pos 24 + accumulators 32 = 56 B  => 137 B (velocity/flag writes happen only for the ~0.06 % of
// particles that hit a wall).  One thread per particle, 256-thread blocks, fully coalesced 8-B lanes.
//
// Legality of the fusion: every wall case and every bounds test reads and writes only the particle itself, and
// the reference evaluates the cases in a fixed order with each mask computed after the previous handler ran
// (Pore:442-485) — which is exactly a per-particle sequential evaluation.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "amc_commit_dev.h"

// One particle through the stages of a streaming pass, in registers: the previous step's post-sweep bounds check (if it
// was left to this pass), drift, the wall cases in reference order, this step's bounds check.  px, py, pz: prior_*_vals —
// set here by a drift stage, supplied by the caller for a walls-only pass.
struct amc_stream_counts {
    int nwall, nerr, noob, noob_pre;
};

template <int GEOM>
AMC_DEV void amc_stream_particle(amc_particle &q, const amc_params &P, const amc_out &O, double dt, int stages, int ip,
                                 double &px, double &py, double &pz, amc_stream_counts &cn)
{
    // Pore:550 of the previous step: nothing happens between that bounds check and this step's drift, and it touches
    // only the particle itself, so inside a multi-step run it rides along with this pass instead of a pass of its own
    if ((stages & AMC_ST_BOUNDS_PRE) && GEOM != AMC_GEOM_CUBE)
        cn.noob_pre = amc_bounds(P, q.x, q.y, q.z, GEOM == AMC_GEOM_PORE_ENERGISED);
    if (stages & AMC_ST_DRIFT) {                  // Pore:427-437 / Cube:180-187 (ndarray ops: exact products)
        px = q.x; py = q.y; pz = q.z;             // prior_*_vals (Pore:427-429)
        const double sx = dt * q.vx, sy = dt * q.vy, sz = dt * q.vz;
        q.x += sx; q.y += sy; q.z += sz;
        q.d += fabs(sqrt(sx * sx + sy * sy + sz * sz));
        q.dx += fabs(sx); q.dy += fabs(sy); q.dz += fabs(sz);
    }
    if (stages & AMC_ST_WALLS) {
        if (GEOM == AMC_GEOM_CUBE) {              // Cube:189-226: per axis, max wall then min wall
#define AMC_CUBE_AXIS(pos, vel, W)                                                                    \
    if (pos > W) { const double t_ = (pos - W) / vel; vel = -vel; pos = W + t_ * vel; }                \
    if (pos < 0) { const double t_ = pos / vel; vel = -vel; pos = t_ * vel; }
            AMC_CUBE_AXIS(q.x, q.vx, P.cube_x)
            AMC_CUBE_AXIS(q.y, q.vy, P.cube_y)
            AMC_CUBE_AXIS(q.z, q.vz, P.cube_z)
#undef AMC_CUBE_AXIS
        } else if (GEOM == AMC_GEOM_PORE) {       // Pore:439-485, cases in order
            const double zb = P.z_gap_bottom, zt = P.z_gap_top;
            const double r0 = sqrt(px * px + py * py);
            if (sqrt(q.x * q.x + q.y * q.y) > P.R_oa) {                                  // CASE 1, Pore:442
                if (amc_side_wall(q, P.R_oa_c, true, O, 1, ip)) cn.nerr++; else cn.nwall++;
            }
            if (q.z < 0) { amc_vertical_wall(q, 0.0, O, 2, ip); cn.nwall++; }           // CASE 2, Pore:448
            if (q.z > P.H) { amc_vertical_wall(q, P.H, O, 3, ip); cn.nwall++; }         // Pore:451
            if ((pz > P.z_cold) && (q.z < P.z_cold) && (sqrt(q.x * q.x + q.y * q.y) > P.R_p)) {   // CASE 3 cold, Pore:457
                amc_vertical_wall(q, P.z_cold, O, 4, ip); cn.nwall++;
            }
            if ((pz < P.h_oa) && (q.z > P.h_oa) && (sqrt(q.x * q.x + q.y * q.y) > P.R_p)) {       // CASE 3 hot, Pore:460
                amc_vertical_wall(q, P.h_oa, O, 5, ip); cn.nwall++;
            }
            if ((pz < zt) && (pz > zb) && (r0 < P.R_g) && (sqrt(q.x * q.x + q.y * q.y) > P.R_g)) { // CASE 4, Pore:465
                if (amc_side_wall(q, P.R_g_c, true, O, 6, ip)) cn.nerr++; else cn.nwall++;
            }
            if ((r0 > P.R_p) && (q.z < zb) && (pz < zt) && (pz > zb)) {                 // CASE 5 bottom, Pore:472
                amc_vertical_wall(q, zb, O, 7, ip); cn.nwall++;
            }
            if ((r0 > P.R_p) && (q.z > zt) && (pz < zt) && (pz > zb)) {                 // CASE 5 top, Pore:476
                amc_vertical_wall(q, zt, O, 8, ip); cn.nwall++;
            }
            if ((r0 < P.R_p) && (sqrt(q.x * q.x + q.y * q.y) > P.R_p) &&
                (((q.z < P.z_cold) && (q.z > zt)) || ((q.z < zb) && (q.z > P.h_oa)))) { // CASE 6, Pore:482
                if (amc_side_wall(q, P.R_p_c, true, O, 9, ip)) cn.nerr++; else cn.nwall++;
            }
        } else if (GEOM == AMC_GEOM_PORE_ENERGISED) {
            // Temp:693-703 — cases 1 and 2 are specular WITHOUT free-path bookkeeping or counter (Temp:311-347);
            // the energised cases 3-6 follow in amc_energised.hip around the host's random draws
            if (sqrt(q.x * q.x + q.y * q.y) > P.R_oa) {                                  // CASE 1, Temp:693
                if (amc_side_wall(q, P.R_oa_c, false, O, 1, ip)) cn.nerr++;
            }
            if (q.z < 0) { const double t_ = (q.z - 0.0) / q.vz; q.vz = -q.vz; q.z = 0.0 + t_ * q.vz; }       // Temp:699
            if (q.z > P.H) { const double t_ = (q.z - P.H) / q.vz; q.vz = -q.vz; q.z = P.H + t_ * q.vz; }       // Temp:702
        }
    }
    if ((stages & AMC_ST_BOUNDS) && GEOM != AMC_GEOM_CUBE)
        cn.noob = amc_bounds(P, q.x, q.y, q.z, GEOM == AMC_GEOM_PORE_ENERGISED);        // Pore:512 / Temp:804
}

// the pass's counters: applied directly, or deferred with its other events (amc_out::wev)
AMC_DEV void amc_stream_count(const amc_out &O, const amc_stream_counts &cn, int bounds_slot, int ip)
{
    if (!(cn.nwall | cn.nerr | cn.noob | cn.noob_pre)) return;
    if (O.wev.count) {
        amc_wev_rec r;
        r.p = ip; r.kind = 1; r.a = cn.nwall | (cn.nerr << 16); r.b = cn.noob | (cn.noob_pre << 16);
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.0;
        amc_wev_append(O, r);
        return;
    }
    if (cn.nwall) atomicAdd(&O.banks[amc_bank_id()].n_wall, (unsigned long long)cn.nwall);
    if (cn.nerr) atomicAdd(&O.banks[amc_bank_id()].n_fp_errors, (unsigned long long)cn.nerr);
    if (cn.noob) atomicAdd(bounds_slot ? &O.cnt->n_oob_pp : &O.cnt->n_oob_walls, (unsigned long long)cn.noob);
    if (cn.noob_pre) atomicAdd(&O.cnt->n_oob_pp, (unsigned long long)cn.noob_pre);
}

// The pass touches every state word exactly once: nontemporal loads and stores keep the 137 MB of a step at N = 1e6 from
// pushing the per-cell list heads and records (re-read at once by the detect kernel, hit by this kernel's own atomics) out
// of L2 / the Infinity Cache.  Same-session A/B, three alternating runs each: k_stream 61.4 -> 55.5 us (pore, N = 1e6),
// 63.2 -> 59.1 us (cube); whole step -4 % / -3 %.
#define AMC_LD(p) __builtin_nontemporal_load(&(p))
#define AMC_ST(p, v) __builtin_nontemporal_store((v), &(p))

template <int GEOM>
__global__ __launch_bounds__(256) void k_stream(amc_state S, amc_state S_out, amc_params P, amc_out O, double dt, int stages,
                                                long long lo, long long hi, int keep_prior, int bounds_slot,
                                                amc_grid G, amc_lists B, int build_lists, amc_lazy L, amc_commit_args C,
                                                amc_ovl V)
{
    // the previous sweep's commit rides along on EXTRA blocks behind the streaming ones (amc_commit_dev.h): order-free work
    // that nothing in this pass depends on (results reach the particles through slot_of[] below), done while the others stream
    if (C.enabled) {
        const unsigned nstream = gridDim.x - (unsigned)C.enabled;       // (enabled = number of commit blocks)
        if (blockIdx.x >= nstream) {
            amc_commit_part(C, O, G, S, (int)((blockIdx.x - nstream) * blockDim.x + threadIdx.x), (int)(C.enabled * blockDim.x));
            return;
        }
    }
    const long long p = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hi) return;
    // an overlapped run: the sweep of the previous step is still being resolved.  Particles its detect kernel
    // linked into a candidate are left to the fix-up kernel, which advances them from the sweep's results.
    if (V.skip_epoch && (unsigned int)(V.adj_head[p] >> 32) == V.skip_epoch) return;      // (epoch 0: no sweep in flight)
    // (kept lists, a step in between: what this wave has handed out of its node pool so far — one word only it touches)
    const int wave_id = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int keep_count0 = build_lists == 3 ? B.wave_count[wave_id] : 0;
    // (and where the particle is filed: asked for now, needed at the very end — there the round trip would be the wave's last)
    const int keep_cell = build_lists == 3 ? B.cell_of[p] : -1, keep_node = build_lists == 3 ? B.node_of[p] : (int)p;
    amc_particle q;
    q.x = AMC_LD(S.x[p]); q.y = AMC_LD(S.y[p]); q.z = AMC_LD(S.z[p]);
    q.vx = AMC_LD(S.vx[p]); q.vy = AMC_LD(S.vy[p]); q.vz = AMC_LD(S.vz[p]);
    const double x_in = q.x, y_in = q.y, z_in = q.z, vx_in = q.vx, vy_in = q.vy, vz_in = q.vz;
    const bool need_acc = (stages & (AMC_ST_DRIFT | AMC_ST_WALLS)) != 0;
    bool flag_in = false;
    double d_in = 0, dx_in = 0, dy_in = 0, dz_in = 0;
    q.d = q.dx = q.dy = q.dz = 0; q.flag = false;
    if (need_acc) {
        q.d = AMC_LD(S.d[p]); q.dx = AMC_LD(S.dx[p]); q.dy = AMC_LD(S.dy[p]); q.dz = AMC_LD(S.dz[p]);
        q.flag = S.flag[p] != 0;
        flag_in = q.flag; d_in = q.d; dx_in = q.dx; dy_in = q.dy; dz_in = q.dz;
    }
    // deferred commit of the previous sweep: a particle that collided has its new state in the slot arrays
    bool force = S_out.x != S.x;        // (another buffer: every field is written)
    if (L.enabled) {
        const int sl = L.slot_of[p];
        if (sl >= 0) {
            L.slot_of[p] = -1;
            if (L.moved[sl]) {
                const double *t = L.state + (size_t)sl * RS_SLOT_DOUBLES;
                q.x = t[0]; q.y = t[1]; q.z = t[2]; q.vx = t[3]; q.vy = t[4]; q.vz = t[5];
                q.d = t[6]; q.dx = t[7]; q.dy = t[8]; q.dz = t[9]; q.flag = t[10] != 0.0;
                force = true;       // the particle arrays are stale: write every field back
            }
        }
    }
    const bool wr_acc = need_acc || force;
    double px = q.x, py = q.y, pz = q.z;
    if (!(stages & AMC_ST_DRIFT) && GEOM != AMC_GEOM_CUBE && (stages & AMC_ST_WALLS)) { px = S.px[p]; py = S.py[p]; pz = S.pz[p]; }
    amc_stream_counts cn = {0, 0, 0, 0};
    amc_stream_particle<GEOM>(q, P, O, dt, stages, (int)p, px, py, pz, cn);
    if ((stages & AMC_ST_DRIFT) && keep_prior && GEOM != AMC_GEOM_CUBE) { S.px[p] = px; S.py[p] = py; S.pz[p] = pz; }

    // write back only what changed (positions and accumulators always change in a drift step)
    if (force || q.x != x_in) AMC_ST(S_out.x[p], q.x);
    if (force || q.y != y_in) AMC_ST(S_out.y[p], q.y);
    if (force || q.z != z_in) AMC_ST(S_out.z[p], q.z);
    if (force || q.vx != vx_in) S_out.vx[p] = q.vx;
    if (force || q.vy != vy_in) S_out.vy[p] = q.vy;
    if (force || q.vz != vz_in) S_out.vz[p] = q.vz;
    if (wr_acc) {
        if (force || q.d != d_in) AMC_ST(S_out.d[p], q.d);
        if (force || q.dx != dx_in) AMC_ST(S_out.dx[p], q.dx);
        if (force || q.dy != dy_in) AMC_ST(S_out.dy[p], q.dy);
        if (force || q.dz != dz_in) AMC_ST(S_out.dz[p], q.dz);
        if (force || q.flag != flag_in) S_out.flag[p] = q.flag ? 1 : 0;
    }
    // fused build of the detection grid's per-cell lists (amc_grid.hip): the particle's final position of this stage
    // is in registers, so no separate binning pass over the positions is needed
    // (build_lists: 1 anew every step; kept lists — 2 the full build of a cycle, which also empties the node pools, 3 a step
    // in between)
    if (build_lists == 1) {
        bool outside = false;
        amc_list_insert(G, B, (int)p, q.x, q.y, q.z, &outside);
        if (outside) atomicOr(&O.cnt->flags, 8ULL);
    } else if (build_lists) {
        bool outside = false, overflow = false;
        const int nc = amc_list_keep(G, B, (int)p, q.x, q.y, q.z, build_lists == 2, wave_id, keep_count0, keep_cell, keep_node, &outside, &overflow);
        if ((int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) B.wave_count[wave_id] = nc;     // (the wave's own word)
        if (outside) atomicOr(&O.cnt->flags, 8ULL);
        if (overflow) atomicOr(&O.cnt->flags, 1ULL);
    }
    amc_stream_count(O, cn, bounds_slot, (int)p);
}

// ---- the fix-up kernel of an overlapped run ---------------------------------------------------------------
// Step s + 1's streaming pass ran while sweep s was being resolved: it advanced every particle that was in no candidate of
// sweep s, from buffer S_in into buffer S_out, filed it in the next lists and DEFERRED its events.  What is left:
//   1. every particle that holds a slot of sweep s: its post-sweep state (the slot's, if the sweep moved it, else S_in's)
//      goes through the SAME streaming stages into S_out.  One that was in a candidate was left out by the pass: it is filed
//      now.  One that the sweep pulled in later (a "victim": in no candidate when the pass started) was advanced speculatively
//      from its pre-sweep state: its node keeps its place in the list it was filed under but gets a position no test passes,
//      and the particle is filed again under an extra node (amc_lists);
//   2. the commit of sweep s (completed paths, counters, overlay lists emptied) — what rides along with the streaming pass in
//      a plain run;
//   3. the deferred events of the pass, except those of victims (redone in 1, with their events applied directly).
struct amc_fixup_args {
    amc_state S_in, S_out;
    amc_lists B;                 // the NEXT lists (being completed here)
    int *extra;                  // its extra nodes: node n + e -> particle, and their count
    int *extra_count;
    int max_extra;
    const unsigned long long *adj_head;
    const unsigned int *victim;
    unsigned int sweep_epoch;    // of sweep s (0: there was none — the first step of a run)
    amc_wev wev;                 // the pass's deferred events ...
    unsigned int *wev_clear;     // ... and the counters of the buffer the previous fix-up consumed (cleared for the pass after next)
    int *extra_clear;            // likewise the extra-node counter of the lists that have been swept
    int stages, step_sweep, step_stream;
    double dt;
};

#define AMC_FIXUP_SLOT_BLOCKS 24     // of the AMC_COMMIT_BLOCKS blocks: the slots; the others: the sweep's commit, side by side

template <int GEOM>
__global__ __launch_bounds__(256) void k_fixup(amc_fixup_args F, amc_params P, amc_out O, amc_grid G, amc_commit_args C)
{
    if (blockIdx.x == 0 && threadIdx.x < AMC_COUNTER_BANKS) F.wev_clear[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *F.extra_clear = 0;
    if (F.sweep_epoch && blockIdx.x < AMC_FIXUP_SLOT_BLOCKS) {
        // 1. the slots (the three parts are independent of each other: they run on different blocks at the same time)
        const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gstride = AMC_FIXUP_SLOT_BLOCKS * blockDim.x;
        amc_resolve_ctl *ctl = C.ctl;
        const bool active = ctl->active != 0;
        const bool ok = ctl->ok && !ctl->ovf;
        const int ns = !active ? 0 : (ctl->nslots < C.max_slots ? ctl->nslots : C.max_slots);
        amc_out Od = O;
        Od.wev.count = nullptr;                     // (events of particles advanced here are final)
        Od.step = F.step_stream;
        for (int s = gtid; s < ns; s += gstride) {
            const int p = C.sl_meta[s].x;
            if (p < 0) continue;                    // a candidate's slot that no particle took
            amc_particle q;
            if (ok && C.sl_moved[s]) {
                const double *t = C.sl_state + (size_t)s * RS_SLOT_DOUBLES;
                q.x = t[0]; q.y = t[1]; q.z = t[2]; q.vx = t[3]; q.vy = t[4]; q.vz = t[5];
                q.d = t[6]; q.dx = t[7]; q.dy = t[8]; q.dz = t[9]; q.flag = t[10] != 0.0;
            } else {
                q.x = F.S_in.x[p]; q.y = F.S_in.y[p]; q.z = F.S_in.z[p]; q.vx = F.S_in.vx[p]; q.vy = F.S_in.vy[p]; q.vz = F.S_in.vz[p];
                q.d = F.S_in.d[p]; q.dx = F.S_in.dx[p]; q.dy = F.S_in.dy[p]; q.dz = F.S_in.dz[p]; q.flag = F.S_in.flag[p] != 0;
            }
            const bool marked = (unsigned int)(F.adj_head[p] >> 32) == F.sweep_epoch;
            double px = q.x, py = q.y, pz = q.z;
            amc_stream_counts cn = {0, 0, 0, 0};
            amc_stream_particle<GEOM>(q, P, Od, F.dt, F.stages, p, px, py, pz, cn);
            F.S_out.x[p] = q.x; F.S_out.y[p] = q.y; F.S_out.z[p] = q.z;
            F.S_out.vx[p] = q.vx; F.S_out.vy[p] = q.vy; F.S_out.vz[p] = q.vz;
            F.S_out.d[p] = q.d; F.S_out.dx[p] = q.dx; F.S_out.dy[p] = q.dy; F.S_out.dz[p] = q.dz; F.S_out.flag[p] = q.flag ? 1 : 0;
            bool outside = false;
            if (marked) {
                amc_list_insert(G, F.B, p, q.x, q.y, q.z, &outside);        // left out by the pass: filed now
            } else {
                // a victim: filed by the pass under its speculative position.  That node stays in its list, unreachable for
                // any distance test; the particle gets an extra node under its true position.
                ((float *)&F.B.rec[p])[0] = __int_as_float(0x7fc00000);
                const int e = atomicAdd(F.extra_count, 1);
                atomicAdd(&O.cnt->n_refiled, 1);
                if (e < F.max_extra) {
                    F.extra[e] = p;
                    amc_list_insert(G, F.B, F.B.n + e, q.x, q.y, q.z, &outside);
                } else {
                    atomicOr(&O.cnt->flags, 4ULL);  // (work-space overflow: the step reports AMC_ERR_CAPACITY)
                }
            }
            if (outside) atomicOr(&O.cnt->flags, 8ULL);
            C.slot_of[p] = -1;
            amc_stream_count(Od, cn, 0, p);
        }
    } else if (F.sweep_epoch) {
        // 2. the sweep's commit (its results stay where they are: the blocks above consume them from the slot arrays)
        amc_out Os = O;
        Os.wev.count = nullptr;
        Os.step = F.step_sweep;
        amc_commit_part(C, Os, G, F.S_in, (int)((blockIdx.x - AMC_FIXUP_SLOT_BLOCKS) * blockDim.x + threadIdx.x),
                        (int)((gridDim.x - AMC_FIXUP_SLOT_BLOCKS) * blockDim.x));
    }
    // 3. the pass's deferred events: bank b by block b (mod the grid) — one counter load per block, the banks side by side
    amc_out Od = O;
    Od.wev.count = nullptr;
    Od.step = F.step_stream;
    for (int b = blockIdx.x; b < AMC_COUNTER_BANKS; b += gridDim.x) {
        unsigned int nb = F.wev.count[b];
        if (nb > (unsigned)F.wev.cap) nb = (unsigned)F.wev.cap;
        for (unsigned int k = threadIdx.x; k < nb; k += blockDim.x) {
            const amc_wev_rec r = F.wev.rec[(size_t)b * F.wev.cap + k];
            if (F.sweep_epoch && F.victim[r.p] == F.sweep_epoch) continue;      // redone above
            if (r.kind == 0) {
                amc_emit(Od, r.a, 0, r.p, -1, 0, r.v[0], r.v[1], r.v[2], r.v[3]);
            } else {
                amc_stream_counts cn = {r.a & 0xffff, r.a >> 16, r.b & 0xffff, r.b >> 16};
                amc_stream_count(Od, cn, 0, r.p);
            }
        }
    }
}

amc_commit_args amc_make_commit_args(amc_ctx *c)
{
    const amc_resolve_ws &W = c->W;
    amc_commit_args C;
    C.ctl = (amc_resolve_ctl *)W.ctl; C.hist = W.hist; C.ov_head = W.ov_head; C.ev_gen = W.ev_gen; C.ev = W.ev;
    C.sl_meta = W.sl_meta; C.sl_hits = W.sl_hits; C.sl_moved = W.sl_moved; C.sl_state = W.sl_state; C.slot_of = W.slot_of;
    C.max_slots = W.max_slots; C.max_hist = W.max_hist; C.lo = c->lo; C.hi = c->hi; C.count_pp = c->mg_count_pp ? 1 : 0;
    C.defer = c->commit_defer ? 1 : 0; C.nogrid = c->allpairs ? 1 : 0; C.enabled = 0;
    return C;
}

hipError_t amc_launch_stream(amc_ctx *c, double dt, int stages, int bounds_slot, bool fuse_bin)
{
    static const int threads = getenv("AMC_STREAM_BS") ? atoi(getenv("AMC_STREAM_BS")) : 256;      // (experiments: 64 / 128 / 256)
    int build = 0;
    if (fuse_bin && !c->allpairs && c->lo == 0 && c->hi == c->n) {
        if (c->keep_K >= 2 && threads == c->keep_threads && c->B.cell_of) {
            // kept lists: a full build when the lists are not this pass's own or the cycle is over, else a step in between
            if (c->lists_owner != 1) c->lists_age = -1;
            c->lists_owner = 1;
            if (c->lists_age < 0 || c->lists_age + 1 >= c->keep_K) { build = 2; c->B.epoch++; c->lists_age = 0; }
            else { build = 3; c->lists_age++; }
        } else {
            build = 1; c->B.epoch++; c->lists_age = -1;
        }
    }
    amc_lazy L;
    memset(&L, 0, sizeof L);
    if (c->lazy_pending) {              // (a shard: its own particles here, the slots of the others are cleared by the unpack)
        const amc_resolve_ws &W = c->W;
        L.slot_of = W.slot_of; L.state = W.sl_state; L.moved = W.sl_moved;
        L.enabled = 1;
        c->lazy_pending = false;        // this pass consumes them
    }
    const long long cnt = c->hi - c->lo;
    if (cnt <= 0) return hipSuccess;
    amc_commit_args C = amc_make_commit_args(c);
    unsigned extra = 0;
    if (c->commit_pending) {            // this launch does the last sweep's commit as well, on blocks of its own
        const long long lag = c->h_host_ncand ? *c->h_host_ncand : 0;      // (entries to commit ~ 2 per candidate)
        extra = (unsigned)std::min<long long>(std::max<long long>((2 * lag + 255) / 256, 4), AMC_COMMIT_BLOCKS);
        C.enabled = (int)extra;
        c->commit_pending = false;
    }
    const unsigned blocks = (unsigned)((cnt + threads - 1) / threads) + extra;
    const int kp = c->keep_prior ? 1 : 0;
    amc_ovl V;
    V.adj_head = nullptr; V.skip_epoch = 0;
    amc_prof_begin(c, (stages == AMC_ST_BOUNDS) ? AMC_K_BOUNDS : AMC_K_DRIFT_WALLS);
    switch (c->P.geometry) {
    case AMC_GEOM_CUBE:
        AMC_LAUNCH(c, k_stream<AMC_GEOM_CUBE>, dim3(blocks), dim3(threads), c->S, c->S, c->P, c->out, dt,
                           stages, c->lo, c->hi, kp, bounds_slot, c->G, c->B, build, L, C, V);
        break;
    case AMC_GEOM_PORE:
        AMC_LAUNCH(c, k_stream<AMC_GEOM_PORE>, dim3(blocks), dim3(threads), c->S, c->S, c->P, c->out, dt,
                           stages, c->lo, c->hi, kp, bounds_slot, c->G, c->B, build, L, C, V);
        break;
    case AMC_GEOM_PORE_ENERGISED:
        AMC_LAUNCH(c, k_stream<AMC_GEOM_PORE_ENERGISED>, dim3(blocks), dim3(threads), c->S, c->S, c->P,
                           c->out, dt, stages, c->lo, c->hi, kp, bounds_slot, c->G, c->B, build, L, C, V);
        break;
    default:
        break;
    }
    amc_prof_end(c);
    return hipGetLastError();
}

// ---- the overlapped run's two launches (whole range in one context, cube / specular pore, binned detector) ---------------------
// `from`: the buffer (0 / 1) holding the state and the lists of the sweep in flight; the pass writes the other one.
hipError_t amc_launch_stream_ovl(amc_ctx *c, double dt, int stages, int from, unsigned int skip_epoch, hipStream_t stream,
                                 bool build_lists)
{
    const int to = 1 - from;
    amc_lists &Bn = c->B_buf[to];
    Bn.epoch++;
    c->lists_age = -1;
    amc_lazy L;
    memset(&L, 0, sizeof L);
    amc_commit_args C = amc_make_commit_args(c);
    amc_ovl V;
    V.adj_head = c->W.adj_head; V.skip_epoch = skip_epoch;
    amc_out O = c->out;
    O.wev = c->wev_buf[c->out.step & 1];
    const unsigned blocks = (unsigned)((c->n + 255) / 256);
    amc_prof_begin(c, AMC_K_DRIFT_WALLS);
    if (c->P.geometry == AMC_GEOM_CUBE)
        AMC_LAUNCH_ON(c, stream, k_stream<AMC_GEOM_CUBE>, dim3(blocks), dim3(256), c->S_buf[from], c->S_buf[to], c->P, O, dt, stages,
                      0LL, (long long)c->n, 0, 0, c->G, Bn, build_lists ? 1 : 0, L, C, V);
    else
        AMC_LAUNCH_ON(c, stream, k_stream<AMC_GEOM_PORE>, dim3(blocks), dim3(256), c->S_buf[from], c->S_buf[to], c->P, O, dt, stages,
                      0LL, (long long)c->n, 0, 0, c->G, Bn, build_lists ? 1 : 0, L, C, V);
    amc_prof_end(c);
    return hipGetLastError();
}

hipError_t amc_launch_fixup(amc_ctx *c, double dt, int stages, int from, unsigned int sweep_epoch)
{
    const int to = 1 - from;
    amc_fixup_args F;
    F.S_in = c->S_buf[from]; F.S_out = c->S_buf[to];
    F.B = c->B_buf[to];
    F.extra = c->extra_buf[to]; F.extra_count = c->extra_count + to; F.max_extra = c->max_extra;
    F.adj_head = c->W.adj_head; F.victim = c->W.victim; F.sweep_epoch = sweep_epoch;
    F.wev = c->wev_buf[c->out.step & 1];
    F.wev_clear = c->wev_buf[1 - (c->out.step & 1)].count;
    F.extra_clear = c->extra_count + from;
    F.stages = stages; F.step_stream = c->out.step; F.step_sweep = c->out.step - 1; F.dt = dt;
    amc_commit_args C = amc_make_commit_args(c);
    C.defer = 1;                        // (the results are consumed from the slot arrays by the kernel itself)
    C.enabled = 1;
    amc_prof_begin(c, AMC_K_FIXUP);
    if (c->P.geometry == AMC_GEOM_CUBE)
        AMC_LAUNCH(c, k_fixup<AMC_GEOM_CUBE>, dim3(AMC_COMMIT_BLOCKS), dim3(256), F, c->P, c->out, c->G, C);
    else
        AMC_LAUNCH(c, k_fixup<AMC_GEOM_PORE>, dim3(AMC_COMMIT_BLOCKS), dim3(256), F, c->P, c->out, c->G, C);
    amc_prof_end(c);
    return hipGetLastError();
}
