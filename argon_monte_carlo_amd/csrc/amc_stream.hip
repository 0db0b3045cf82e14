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
#include <string.h>

#include <algorithm>

#include "amc_commit_dev.h"

template <int GEOM>
__global__ __launch_bounds__(256) void k_stream(amc_state S, amc_params P, amc_out O, double dt, int stages,
                                                long long lo, long long hi, int keep_prior, int bounds_slot,
                                                amc_grid G, amc_lists B, int build_lists, amc_lazy L, amc_commit_args C)
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
    amc_particle q;
    q.x = S.x[p]; q.y = S.y[p]; q.z = S.z[p];
    q.vx = S.vx[p]; q.vy = S.vy[p]; q.vz = S.vz[p];
    const double x_in = q.x, y_in = q.y, z_in = q.z, vx_in = q.vx, vy_in = q.vy, vz_in = q.vz;
    const bool need_acc = (stages & (AMC_ST_DRIFT | AMC_ST_WALLS)) != 0;
    bool flag_in = false;
    double d_in = 0, dx_in = 0, dy_in = 0, dz_in = 0;
    if (need_acc) {
        q.d = S.d[p]; q.dx = S.dx[p]; q.dy = S.dy[p]; q.dz = S.dz[p];
        q.flag = S.flag[p] != 0;
        flag_in = q.flag; d_in = q.d; dx_in = q.dx; dy_in = q.dy; dz_in = q.dz;
    }
    // deferred commit of the previous sweep: a particle that collided has its new state in the slot arrays
    bool force = false;
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
    // Pore:550 of the previous step: nothing happens between that bounds check and this step's drift, and it touches
    // only the particle itself, so inside a multi-step run it rides along with this pass instead of a pass of its own
    int noob_pre = 0;
    if ((stages & AMC_ST_BOUNDS_PRE) && GEOM != AMC_GEOM_CUBE)
        noob_pre = amc_bounds(P, q.x, q.y, q.z, GEOM == AMC_GEOM_PORE_ENERGISED);
    double px = q.x, py = q.y, pz = q.z;          // prior_*_vals (Pore:427-429)
    const bool wr_acc = need_acc || force;

    if (stages & AMC_ST_DRIFT) {                  // Pore:430-437 / Cube:180-187 (ndarray ops: exact products)
        const double sx = dt * q.vx, sy = dt * q.vy, sz = dt * q.vz;
        q.x += sx; q.y += sy; q.z += sz;
        q.d += fabs(sqrt(sx * sx + sy * sy + sz * sz));
        q.dx += fabs(sx); q.dy += fabs(sy); q.dz += fabs(sz);
        if (keep_prior && GEOM != AMC_GEOM_CUBE) { S.px[p] = px; S.py[p] = py; S.pz[p] = pz; }
    } else if (GEOM != AMC_GEOM_CUBE && (stages & AMC_ST_WALLS)) {
        px = S.px[p]; py = S.py[p]; pz = S.pz[p];
    }

    int nwall = 0, nerr = 0;
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
            const int ip = (int)p;
            const double zb = P.z_gap_bottom, zt = P.z_gap_top;
            const double r0 = sqrt(px * px + py * py);
            if (sqrt(q.x * q.x + q.y * q.y) > P.R_oa) {                                  // CASE 1, Pore:442
                if (amc_side_wall(q, P.R_oa_c, true, O, 1, ip)) nerr++; else nwall++;
            }
            if (q.z < 0) { amc_vertical_wall(q, 0.0, O, 2, ip); nwall++; }              // CASE 2, Pore:448
            if (q.z > P.H) { amc_vertical_wall(q, P.H, O, 3, ip); nwall++; }            // Pore:451
            if ((pz > P.z_cold) && (q.z < P.z_cold) && (sqrt(q.x * q.x + q.y * q.y) > P.R_p)) {   // CASE 3 cold, Pore:457
                amc_vertical_wall(q, P.z_cold, O, 4, ip); nwall++;
            }
            if ((pz < P.h_oa) && (q.z > P.h_oa) && (sqrt(q.x * q.x + q.y * q.y) > P.R_p)) {       // CASE 3 hot, Pore:460
                amc_vertical_wall(q, P.h_oa, O, 5, ip); nwall++;
            }
            if ((pz < zt) && (pz > zb) && (r0 < P.R_g) && (sqrt(q.x * q.x + q.y * q.y) > P.R_g)) { // CASE 4, Pore:465
                if (amc_side_wall(q, P.R_g_c, true, O, 6, ip)) nerr++; else nwall++;
            }
            if ((r0 > P.R_p) && (q.z < zb) && (pz < zt) && (pz > zb)) {                 // CASE 5 bottom, Pore:472
                amc_vertical_wall(q, zb, O, 7, ip); nwall++;
            }
            if ((r0 > P.R_p) && (q.z > zt) && (pz < zt) && (pz > zb)) {                 // CASE 5 top, Pore:476
                amc_vertical_wall(q, zt, O, 8, ip); nwall++;
            }
            if ((r0 < P.R_p) && (sqrt(q.x * q.x + q.y * q.y) > P.R_p) &&
                (((q.z < P.z_cold) && (q.z > zt)) || ((q.z < zb) && (q.z > P.h_oa)))) { // CASE 6, Pore:482
                if (amc_side_wall(q, P.R_p_c, true, O, 9, ip)) nerr++; else nwall++;
            }
        } else if (GEOM == AMC_GEOM_PORE_ENERGISED) {
            // Temp:693-703 — cases 1 and 2 are specular WITHOUT free-path bookkeeping or counter (Temp:311-347);
            // the energised cases 3-6 follow in amc_energised.hip around the host's random draws
            if (sqrt(q.x * q.x + q.y * q.y) > P.R_oa) {                                  // CASE 1, Temp:693
                if (amc_side_wall(q, P.R_oa_c, false, O, 1, (int)p)) nerr++;
            }
            if (q.z < 0) { const double t_ = (q.z - 0.0) / q.vz; q.vz = -q.vz; q.z = 0.0 + t_ * q.vz; }       // Temp:699
            if (q.z > P.H) { const double t_ = (q.z - P.H) / q.vz; q.vz = -q.vz; q.z = P.H + t_ * q.vz; }       // Temp:702
        }
    }

    int noob = 0;
    if ((stages & AMC_ST_BOUNDS) && GEOM != AMC_GEOM_CUBE)
        noob = amc_bounds(P, q.x, q.y, q.z, GEOM == AMC_GEOM_PORE_ENERGISED);        // Pore:512 / Temp:804

    // write back only what changed (positions and accumulators always change in a drift step)
    if (force || q.x != x_in) S.x[p] = q.x;
    if (force || q.y != y_in) S.y[p] = q.y;
    if (force || q.z != z_in) S.z[p] = q.z;
    if (force || q.vx != vx_in) S.vx[p] = q.vx;
    if (force || q.vy != vy_in) S.vy[p] = q.vy;
    if (force || q.vz != vz_in) S.vz[p] = q.vz;
    if (wr_acc) {
        if (force || q.d != d_in) S.d[p] = q.d;
        if (force || q.dx != dx_in) S.dx[p] = q.dx;
        if (force || q.dy != dy_in) S.dy[p] = q.dy;
        if (force || q.dz != dz_in) S.dz[p] = q.dz;
        if (force || q.flag != flag_in) S.flag[p] = q.flag ? 1 : 0;
    }
    // fused build of the detection grid's per-cell lists (amc_grid.hip): the particle's final position of this stage
    // is in registers, so no separate binning pass over the positions is needed
    if (build_lists) {
        bool outside = false;
        amc_list_insert(G, B, (int)p, q.x, q.y, q.z, &outside);
        if (outside) atomicOr(&O.cnt->flags, 8ULL);
    }
    if (nwall) atomicAdd(&O.banks[amc_bank_id()].n_wall, (unsigned long long)nwall);
    if (nerr) atomicAdd(&O.banks[amc_bank_id()].n_fp_errors, (unsigned long long)nerr);
    if (noob) atomicAdd(bounds_slot ? &O.cnt->n_oob_pp : &O.cnt->n_oob_walls, (unsigned long long)noob);
    if (noob_pre) atomicAdd(&O.cnt->n_oob_pp, (unsigned long long)noob_pre);
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
    int build = 0;
    if (fuse_bin && !c->allpairs && c->lo == 0 && c->hi == c->n) { build = 1; c->B.epoch++; }
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
    const int threads = 256;
    const unsigned blocks = (unsigned)((cnt + threads - 1) / threads) + extra;
    const int kp = c->keep_prior ? 1 : 0;
    amc_prof_begin(c, (stages == AMC_ST_BOUNDS) ? AMC_K_BOUNDS : AMC_K_DRIFT_WALLS);
    switch (c->P.geometry) {
    case AMC_GEOM_CUBE:
        AMC_LAUNCH(c, k_stream<AMC_GEOM_CUBE>, dim3(blocks), dim3(threads), c->S, c->P, c->out, dt,
                           stages, c->lo, c->hi, kp, bounds_slot, c->G, c->B, build, L, C);
        break;
    case AMC_GEOM_PORE:
        AMC_LAUNCH(c, k_stream<AMC_GEOM_PORE>, dim3(blocks), dim3(threads), c->S, c->P, c->out, dt,
                           stages, c->lo, c->hi, kp, bounds_slot, c->G, c->B, build, L, C);
        break;
    case AMC_GEOM_PORE_ENERGISED:
        AMC_LAUNCH(c, k_stream<AMC_GEOM_PORE_ENERGISED>, dim3(blocks), dim3(threads), c->S, c->P,
                           c->out, dt, stages, c->lo, c->hi, kp, bounds_slot, c->G, c->B, build, L, C);
        break;
    default:
        break;
    }
    amc_prof_end(c);
    return hipGetLastError();
}
