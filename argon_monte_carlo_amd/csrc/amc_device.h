// amc_device.h — device-side arithmetic of the hot path (gfx950).  Every function cites the reference lines it
// follows (Pore = Open_Air_Pore_MC.py, Cube = Open_Air_Cube_MC.py, Temp = Temperature_Pore_MC.py).
//
// Numerics contract (DESIGN.md "numerics"): IEEE double throughout, compiled with -ffp-contract=off so that the
// only fused multiply-adds are the explicit fma() calls that restate OpenBLAS' ddot tail (np.dot, Pore:209,320);
// fp64 sqrt and divide are the correctly rounded forms.  Squares are exact products (x*x): the reference's NumPy
// *scalar* `x**2` calls libm pow, which differs from x*x by 1 ulp for ~0.08 % of inputs — the CPU oracle has both
// variants (orc_pow_* pinned to the reference, orc_mul_* identical to this file).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/argonmc.h"

#define AMC_DEV __device__ __forceinline__

// device-global work counters (one struct in device memory per ctx)
struct amc_dev_counters {
    unsigned long long n_pp, n_wall, n_oob_walls, n_oob_pp, n_paths, n_candidates, n_clusters, n_rounds, n_fp_errors,
        flags;
    unsigned long long n_paths_total;   // all paths ever emitted (histogram population incl. out-of-range)
    unsigned int path_count;            // records currently in the path buffer
    unsigned int cand_count;            // candidate pairs of the current sweep
    int step;                           // current step index (for record keys)
    int n_refiled;                      // overlapped runs: particles advanced again from a sweep's result and filed under an extra node
};

// Per-event counters are BANKED: returning or not, atomics on one word from different waves complete one every ~12 ns
// (tools/ubench_sameaddr.hip), which at ~10^3 wall hits per step is a serial chain of tens of microseconds inside a
// streaming kernel.  Each wave adds to the bank of its (block, wave) id, one 64-byte line per bank; read_counters()
// folds the banks into amc_dev_counters on the host.
#define AMC_COUNTER_BANKS 64
struct amc_counter_bank {
    unsigned long long n_wall, n_paths, n_paths_total, n_fp_errors, n_pp, pad[3];
};
// (readfirstlane: the id is the same for all lanes of a wave; telling the compiler so keeps its wave-level combining of
// atomics on a uniform address, which otherwise turns one atomic per wave into one per lane)
AMC_DEV int amc_bank_id()
{
    return __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 8u + (threadIdx.x >> 6)) & (AMC_COUNTER_BANKS - 1)));
}

// Deferred events of a streaming pass that runs AHEAD of the previous sweep's resolve (the overlapped run, amc_stream.hip):
// a particle that the sweep pulls into a cluster after the pass has already advanced it is advanced again from its
// collision result, so what the first, speculative pass emitted for it (wall paths, counters) must not count.  The pass
// therefore appends its events here and the fix-up kernel commits them once it knows which particles were redone.
// Banked (one counter word per (block, wave) bank: same-address atomics from many waves are a serial chain).
struct amc_wev_rec {
    int p, kind;                        // particle; 0 = completed path (a = phase), 1 = counters
    int a, b;                           // kind 1: a = wall hits | failed solves << 16, b = out-of-bounds | previous step's << 16
    double v[4];                        // kind 0: total, x, y, z
};
struct amc_wev {
    amc_wev_rec *rec;                   // [AMC_COUNTER_BANKS][cap]
    unsigned int *count;                // [AMC_COUNTER_BANKS]  (nullptr: events are applied directly)
    int cap;
};

// where completed paths go: a record buffer (optional) + the four np.histogram-compatible histograms
struct amc_out {
    amc_path_record *rec;               // nullptr: no records, histograms only
    unsigned int cap;
    amc_counter_bank *banks;            // [AMC_COUNTER_BANKS]
    unsigned long long *hist;           // [AMC_COUNTER_BANKS][4][nbins], summed over the banks on the host
    const double *edges;                // [nbins+1] = np.linspace(lo, hi, nbins+1) (kept for reference; recomputed on the fly)
    double bin_step;                    // (hi - lo) / nbins: edge k = lo + k * bin_step, edge nbins = hi — np.linspace's formula
    int nbins;
    double lo, hi;
    amc_dev_counters *cnt;
    int step;                           // index of the current step (record key), set by the host per launch
    amc_wev wev;                        // count != nullptr: this pass's events are deferred (see amc_wev)
};

AMC_DEV void amc_wev_append(const amc_out &o, const amc_wev_rec &r)
{
    const int bank = amc_bank_id();
    const unsigned int k = atomicAdd(&o.wev.count[bank], 1u);
    if (k < (unsigned)o.wev.cap) o.wev.rec[(size_t)bank * o.wev.cap + k] = r;
    else atomicOr(&o.cnt->flags, 4ULL);         // (work-space overflow: the step reports AMC_ERR_CAPACITY)
}

// np.histogram(a, bins=n, range=(lo,hi)) bin of one value (numpy/lib/_histograms_impl.py uniform-bin path):
// returns -1 if outside [lo, hi].
AMC_DEV int amc_hist_bin(const amc_out &o, double v)
{
    if (!(v >= o.lo && v <= o.hi)) return -1;
    double f = ((v - o.lo) / (o.hi - o.lo)) * (double)o.nbins;
    int idx = (int)f;
    if (idx == o.nbins) idx -= 1;
    // bin edges recomputed instead of loaded (saves two dependent memory round trips per value)
    const double e0 = o.lo + (double)idx * o.bin_step;
    const double e1 = (idx + 1 == o.nbins) ? o.hi : o.lo + (double)(idx + 1) * o.bin_step;
    if (v < e0) idx -= 1;
    else if (v >= e1 && idx != o.nbins - 1) idx += 1;
    return idx;
}

AMC_DEV void amc_emit(const amc_out &o, int phase, long long cell, int i, int j, int which, double tot, double px,
                      double py, double pz)
{
    if (o.wev.count) {                  // a streaming pass ahead of the resolve: deferred (wall events: cell 0, j -1, which 0)
        amc_wev_rec r;
        r.p = i; r.kind = 0; r.a = phase; r.b = 0; r.v[0] = tot; r.v[1] = px; r.v[2] = py; r.v[3] = pz;
        amc_wev_append(o, r);
        return;
    }
    amc_counter_bank &bank = o.banks[amc_bank_id()];
    atomicAdd(&bank.n_paths, 1ULL);
    atomicAdd(&bank.n_paths_total, 1ULL);
    if (o.hist) {
        // free paths pile up in a few low bins: one copy of the histograms per counter bank keeps the chains of
        // same-address atomics short (they cost 15 of the wide commit kernel's 24 us at N = 1e6 otherwise)
        unsigned long long *hb = o.hist + (size_t)amc_bank_id() * 4 * (size_t)o.nbins;
        int b;
        if ((b = amc_hist_bin(o, tot)) >= 0) atomicAdd(&hb[0 * o.nbins + b], 1ULL);
        if ((b = amc_hist_bin(o, px)) >= 0) atomicAdd(&hb[1 * o.nbins + b], 1ULL);
        if ((b = amc_hist_bin(o, py)) >= 0) atomicAdd(&hb[2 * o.nbins + b], 1ULL);
        if ((b = amc_hist_bin(o, pz)) >= 0) atomicAdd(&hb[3 * o.nbins + b], 1ULL);
    }
    if (o.rec) {
        unsigned int k = atomicAdd(&o.cnt->path_count, 1u);
        if (k < o.cap) {
            amc_path_record r;
            r.step = o.step; r.phase = phase; r.cell = cell; r.i = i; r.j = j; r.which = which; r.reserved = 0;
            r.total = tot; r.px = px; r.py = py; r.pz = pz;
            o.rec[k] = r;
        } else {
            atomicOr(&o.cnt->flags, 2ULL);
        }
    }
}

// One particle's state in registers.
struct amc_particle {
    double x, y, z, vx, vy, vz, d, dx, dy, dz;
    bool flag;
};

// fp64 divide / sqrt are ~40-instruction expansions (correctly rounded).  The streaming kernel inlines them; the
// resolve kernels (tens of KB of code executed once per launch by a handful of waves, i.e. instruction-fetch bound)
// define AMC_COMPACT_MATH and call one shared copy instead.  Same arithmetic either way.
#ifdef AMC_COMPACT_MATH
static __device__ __noinline__ double amc_div(double a, double b) { return a / b; }
static __device__ __noinline__ double amc_sqrt(double a) { return sqrt(a); }
#else
AMC_DEV double amc_div(double a, double b) { return a / b; }
AMC_DEV double amc_sqrt(double a) { return sqrt(a); }
#endif

AMC_DEV double amc_speed(double vx, double vy, double vz) { return amc_sqrt(vx * vx + vy * vy + vz * vz); }

// Pore:173-174 — the overlap test
AMC_DEV bool amc_overlap(double x1, double y1, double z1, double x2, double y2, double z2, double cr)
{
    double ex = x2 - x1, ey = y2 - y1, ez = z2 - z1;
    return amc_sqrt(ex * ex + ey * ey + ez * ez) < cr;
}

// Pore:176-241 — resolve one detected collision between p1 (= j, lower rank) and p2 (= i).
// Returns 0, or 1 when the reference would raise FloatingPointError (a == 0 / negative discriminant).
// emit(which, tot, px, py, pz) is called for each completed free path, j first then i (Pore:186-199).
template <class Emit>
AMC_DEV int amc_collide(amc_particle &p1, amc_particle &p2, double cr, double m, Emit emit)
{
    const double x1 = p1.x, x2 = p2.x, y1 = p1.y, y2 = p2.y, z1 = p1.z, z2 = p2.z;
    const double vx1 = p1.vx, vx2 = p2.vx, vy1 = p1.vy, vy2 = p2.vy, vz1 = p1.vz, vz2 = p2.vz;
    const double ex = x2 - x1, ey = y2 - y1, ez = z2 - z1;
    const double ux = -vx2 + vx1, uy = -vy2 + vy1, uz = -vz2 + vz1;
    const double a = ux * ux + uy * uy + uz * uz;                                   // Pore:182
    const double b = 2 * (ex * ux + ey * uy + ez * uz);                              // Pore:183
    const double c = ex * ex + ey * ey + ez * ez - cr * cr;                          // Pore:184
    const double disc2 = b * b - 4 * a * c;
    if (a == 0.0 || disc2 < 0.0 || a != a || disc2 != disc2) return 1;
    const double sq = amc_sqrt(disc2);
    const double t1 = amc_div(-b + sq, 2 * a), t2 = amc_div(-b - sq, 2 * a);
    const double t = (t1 > t2) ? t1 : t2;                                            // Pore:185
    if (p1.flag)                                                                     // Pore:186-190
        emit(0, fabs(p1.d - fabs(amc_speed(vx1, vy1, vz1) * t)), fabs(p1.dx - fabs(vx1 * t)),
             fabs(p1.dy - fabs(vy1 * t)), fabs(p1.dz - fabs(vz1 * t)));
    else
        p1.flag = true;                                                              // Pore:192
    if (p2.flag)                                                                     // Pore:193-197
        emit(1, fabs(p2.d - fabs(amc_speed(vx2, vy2, vz2) * t)), fabs(p2.dx - fabs(vx2 * t)),
             fabs(p2.dy - fabs(vy2 * t)), fabs(p2.dz - fabs(vz2 * t)));
    else
        p2.flag = true;                                                              // Pore:199
    const double nx1 = x1 - vx1 * t, ny1 = y1 - vy1 * t, nz1 = z1 - vz1 * t;         // Pore:202
    const double nx2 = x2 - vx2 * t, ny2 = y2 - vy2 * t, nz2 = z2 - vz2 * t;
    const double n0 = amc_div(nx2 - nx1, cr), n1 = amc_div(ny2 - ny1, cr), n2 = amc_div(nz2 - nz1, cr);  // Pore:205-207
    const double d1 = fma(vz1, n2, fma(vy1, n1, vx1 * n0));                          // Pore:209 (np.dot = FMA chain)
    const double d2 = fma(vz2, n2, fma(vy2, n1, vx2 * n0));
    const double p = amc_div(d1 - d2, m);
    const double pm = p * m;
    const double wvx1 = vx1 - pm * n0, wvy1 = vy1 - pm * n1, wvz1 = vz1 - pm * n2;   // Pore:211-213
    const double wvx2 = vx2 + pm * n0, wvy2 = vy2 + pm * n1, wvz2 = vz2 + pm * n2;   // Pore:214-216
    p1.x = nx1 + wvx1 * t; p1.y = ny1 + wvy1 * t; p1.z = nz1 + wvz1 * t;            // Pore:218
    p2.x = nx2 + wvx2 * t; p2.y = ny2 + wvy2 * t; p2.z = nz2 + wvz2 * t;            // Pore:219
    p1.vx = wvx1; p1.vy = wvy1; p1.vz = wvz1;                                        // Pore:227-229
    p2.vx = wvx2; p2.vy = wvy2; p2.vz = wvz2;                                        // Pore:230-232
    p2.d = fabs(amc_speed(wvx2, wvy2, wvz2) * t);                                    // Pore:233
    p1.d = fabs(amc_speed(wvx1, wvy1, wvz1) * t);                                    // Pore:234
    p2.dx = fabs(wvx2 * t); p2.dz = fabs(wvz2 * t); p2.dy = fabs(wvy2 * t);          // Pore:235-237
    p1.dx = fabs(wvx1 * t); p1.dy = fabs(wvy1 * t); p1.dz = fabs(wvz1 * t);          // Pore:238-240
    return 0;
}

// Pore:257-292 hit_vertical_wall for one particle.
AMC_DEV void amc_vertical_wall(amc_particle &q, double z_plane, const amc_out &o, int phase, int idx)
{
    const double t = amc_div(q.z - z_plane, q.vz);                                   // Pore:261
    const double sp = amc_speed(q.vx, q.vy, q.vz);
    if (q.flag)                                                                      // Pore:274-278
        amc_emit(o, phase, 0, idx, -1, 0, fabs(q.d - fabs(sp * t)), fabs(q.dx - fabs(q.vx * t)),
                 fabs(q.dy - fabs(q.vy * t)), fabs(q.dz - fabs(q.vz * t)));
    else
        q.flag = true;
    q.d = fabs(sp * t);                                                              // Pore:281
    q.dx = fabs(q.vx * t); q.dy = fabs(q.vy * t); q.dz = fabs(q.vz * t);             // Pore:282-284
    q.vz = -q.vz;                                                                    // Pore:290
    q.z = z_plane + t * q.vz;                                                        // Pore:291
}

// Pore:294-348 hit_cylinder_side_wall for one particle (bookkeeping=false: Temp:317-347).
// Returns 1 when the reference's try-block would fail (no real root).
AMC_DEV int amc_side_wall(amc_particle &q, double Rc, bool bookkeeping, const amc_out &o, int phase, int idx)
{
    const double x = q.x, y = q.y, vx = q.vx, vy = q.vy, vz = q.vz;
    const double a = (-vx) * (-vx) + (-vy) * (-vy);                                  // Pore:312
    const double b = 2 * (x * (-vx) + y * (-vy));                                    // Pore:313
    const double c = x * x + y * y - Rc * Rc;                                        // Pore:314
    const double disc2 = b * b - 4 * a * c;
    if (a == 0.0 || disc2 < 0.0 || disc2 != disc2) return 1;
    const double sq = amc_sqrt(disc2);
    const double t1 = amc_div(-b + sq, 2 * a), t2 = amc_div(-b - sq, 2 * a);
    const double t = (t1 < t2) ? t1 : t2;                                            // Pore:315
    const double cx = x - vx * t, cy = y - vy * t;                                   // Pore:316
    const double n0 = amc_div(cx, Rc), n1 = amc_div(cy, Rc);                         // Pore:318
    const double scalar = fma(vy, n1, vx * n0);                                      // Pore:320 (np.dot)
    const double s2 = 2 * scalar;
    const double wvx = vx - s2 * n0, wvy = vy - s2 * n1;                             // Pore:321
    const double wx = cx + wvx * t, wy = cy + wvy * t;                               // Pore:323
    if (bookkeeping) {
        if (q.flag)                                                                  // Pore:324-328
            amc_emit(o, phase, 0, idx, -1, 0, fabs(q.d - fabs(amc_speed(vx, vy, vz) * t)), fabs(q.dx - fabs(vx * t)),
                     fabs(q.dy - fabs(vy * t)), fabs(q.dz - fabs(vz * t)));
        else
            q.flag = true;
        q.d = fabs(amc_speed(wvx, wvy, vz) * t);                                     // Pore:332
        q.dx = fabs(wvx * t); q.dy = fabs(wvy * t); q.dz = fabs(vz * t);             // Pore:333-335
    }
    q.x = wx; q.y = wy; q.vx = wvx; q.vy = wvy;                                      // Pore:331
    return 0;
}

// Pore:354-375 num_out_of_bounds (mutating) / Temp:594-616 recapture_out_of_bounds, for one particle.
AMC_DEV int amc_bounds(const amc_params &P, double &x, double &y, double &z, bool energised)
{
    int cnt = 0;
    if (z < 0) { if (energised) z = P.oob_z_lo_fix; else z += P.oob_z_lo_fix; cnt++; }
    if (z > P.H) { if (energised) z = P.oob_z_hi_fix; else z -= P.oob_z_hi_fix; cnt++; }
    if (x * x + y * y > P.R_oa_sq) { x = 0; y = 0; cnt++; }
    if ((x * x + y * y > P.R_g_sq) && (z > P.h_oa) && (z < P.z_cold)) { x = 0; y = 0; cnt++; }
    if ((x * x + y * y > P.R_p_sq) &&
        (((z > P.h_oa) && (z < P.z_oob_hot_top)) || ((z > P.z_oob_gap_top) && (z < P.z_cold)))) { x = 0; y = 0; cnt++; }
    return cnt;
}

// Reference-cell membership along one axis for colour group `grp` (Pore:527-529):
//   layer l in [0,nlayers) with ((2l+grp-offset)*d - ov) < v  &&  v < ((2l+grp-offset+1)*d); -1 if none.
AMC_DEV int amc_axis_cell(double v, int grp, int nlayers, int offset, double d, double ov)
{
    const double f = floor(v / d);
    if (!(f > -1.0e9 && f < 1.0e9)) return -1;
    for (int dk = -1; dk <= 1; dk++) {
        const long k = (long)f + dk;
        const long twol = k - grp + offset;
        if (twol < 0 || (twol & 1)) continue;
        const long l = twol / 2;
        if (l >= nlayers) continue;
        const double lo = (double)k * d - ov, hi = (double)(k + 1) * d;
        if (lo < v && v < hi) return (int)l;
    }
    return -1;
}
