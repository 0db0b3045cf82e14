// amc_energised.hip — the deterministic part of Temperature_Pore_MC.py's energised wall handlers (Temp:349-553) on
// the GPU, split around the host-side random draws (Temp:119-141) and mpmath energies (Temp:143-152):
//
//   k_temp_hits   evaluates the mask of ONE case (Temp:708-751) on the current state, and for every hit computes the
//                 flight time since contact, the contact point and the inward unit normal (compacted records);
//   (host)        sorts the records by particle index, draws a direction per hit from the two Mersenne Twisters in that
//                 order, evaluates the surface energy;
//   k_temp_apply  energy accommodation, new velocity, free-path bookkeeping, particle parked at the contact point,
//                 per-hit z-momentum and energy changes (summed by the host in hit order, like Temp:385-389).
//
// The cases are sequential in the reference (each mask is evaluated after the previous handler ran), hence one
// hits/apply pair per case.  Both kernels are O(N) streaming passes over positions (+prior); hits are ~1e-3 of N.
#include "amc_internal.h"
#include "amc_philox.h"

struct temp_records {
    int *idx;
    double *t, *contact, *normal;   // [cap], [3*cap], [3*cap]
    unsigned char *ok;
    int *count;
    int cap;
};

__device__ inline bool temp_mask(const amc_params &P, int case_id, double x, double y, double z, double px, double py,
                                 double pz)
{
    const double r2 = x * x + y * y, r02 = px * px + py * py;
    switch (case_id) {
    case 3: return (pz >= P.t_z3_cold) && (z < P.t_z3_cold) && (r2 > P.R_p_sq);                              // Temp:708
    case 4: return (pz <= P.t_z3_hot) && (z > P.t_z3_hot) && (r2 > P.R_p_sq);                                // Temp:713
    case 5: return (pz < P.t_zgap_hi) && (pz > P.t_zgap_lo) && (r02 <= P.R_g_c_sq) && (r2 > P.R_g_c_sq);     // Temp:720
    case 6: return (r02 >= P.R_p_c_sq) && (z < P.t_zgap_lo) && (pz <= P.t_zgap_hi) && (pz >= P.t_zgap_lo);   // Temp:728
    case 7: return (r02 >= P.R_p_c_sq) && (z > P.t_zgap_hi) && (pz <= P.t_zgap_hi) && (pz >= P.t_zgap_lo);   // Temp:734
    case 8: return (r02 <= P.R_p_c_sq) && (r2 > P.R_p_c_sq) && (z <= P.t_zgap_lo) && (z >= P.t_z3_hot);      // Temp:743
    case 9: return (r02 <= P.R_p_c_sq) && (r2 > P.R_p_c_sq) && (z < P.t_z3_cold) && (z > P.t_zgap_hi);       // Temp:749
    default: return false;
    }
}

// contact of a hit of case `case_id`: flight time since contact, contact point, inward unit normal (Temp:349-375 for the
// planes, Temp:430-474 for the cylinders); ok = 0 when the cylinder solve has no real root (Temp:472-474)
struct temp_contact {
    double t, cx, cy, cz, n0, n1, n2;
    unsigned char ok;
};
__device__ inline temp_contact temp_solve(const amc_params &P, int case_id, double x, double y, double z, double vx,
                                          double vy, double vz)
{
    temp_contact c;
    c.ok = 1; c.t = 0; c.cx = 0; c.cy = 0; c.cz = 0; c.n0 = 0; c.n1 = 0; c.n2 = 0;
    if (case_id == 3 || case_id == 4 || case_id == 6 || case_id == 7) {
        const double zp = case_id == 3 ? P.t_z3_cold : case_id == 4 ? P.t_z3_hot : case_id == 6 ? P.t_zgap_lo : P.t_zgap_hi;
        c.t = (z - zp) / vz;                                                                 // Temp:353
        c.cx = x - vx * c.t; c.cy = y - vy * c.t; c.cz = zp;                                 // Temp:372
        c.n2 = (case_id == 3 || case_id == 6) ? 1.0 : -1.0;                                  // Temp:709,714,730,736
    } else {
        const double Rc = case_id == 5 ? P.R_g_c : P.R_p_c;
        const double a = (-vx) * (-vx) + (-vy) * (-vy);                                      // Temp:436
        const double b = 2 * (x * (-vx) + y * (-vy));
        const double cc = x * x + y * y - Rc * Rc;
        const double disc2 = b * b - 4 * a * cc;
        if (a == 0.0 || disc2 < 0.0 || disc2 != disc2) {
            c.ok = 0;                                                                        // Temp:472-474
        } else {
            const double sq = sqrt(disc2);
            const double t1 = (-b + sq) / (2 * a), t2 = (-b - sq) / (2 * a);
            c.t = (t1 < t2) ? t1 : t2;                                                       // Temp:439
            c.cx = x - vx * c.t; c.cy = y - vy * c.t; c.cz = z - vz * c.t;                   // Temp:440
            c.n0 = -(c.cx / Rc); c.n1 = -(c.cy / Rc); c.n2 = -(0.0 / Rc);                    // Temp:442-444 (negated)
        }
    }
    return c;
}

// energy accommodation and new velocity of a hit (Temp:377-388); returns the particle's speed before the hit
__device__ inline double temp_accommodate(const amc_params &P, int case_id, double vx, double vy, double vz, double Es,
                                          double d0, double d1, double d2, double &wvx, double &wvy, double &wvz,
                                          double &dpz, double &dE)
{
    const double m = P.argon_mass;
    const double alpha = (case_id == 5) ? P.alpha_gap : P.alpha_coated;
    const double v_magnitude = sqrt(vx * vx + vy * vy + vz * vz);                            // Temp:377
    const double old_pz = m * vz;                                                            // Temp:378
    const double E = 0.5 * m * (v_magnitude * v_magnitude);                                  // Temp:128-129,379
    const double diff = Es - E;                                                              // Temp:380
    const double Enew = E + diff * alpha;                                                    // Temp:381
    const double mag = sqrt(Enew * 2 / m);                                                   // Temp:383
    dE = Enew - E;                                                                           // Temp:384
    wvx = d0 * mag; wvy = d1 * mag; wvz = d2 * mag;                                          // Temp:386
    dpz = m * wvz - old_pz;                                                                  // Temp:387-388
    return v_magnitude;
}

__global__ __launch_bounds__(256) void k_temp_hits(amc_state S, amc_params P, int case_id, long long lo, long long hi,
                                                   temp_records R, amc_dev_counters *cnt)
{
    const long long p = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hi) return;
    const double x = S.x[p], y = S.y[p], z = S.z[p];
    if (!temp_mask(P, case_id, x, y, z, S.px[p], S.py[p], S.pz[p])) return;
    const int k = atomicAdd(R.count, 1);
    if (k >= R.cap) { atomicOr(&cnt->flags, 4ULL); return; }
    R.idx[k] = (int)p;
    const temp_contact c = temp_solve(P, case_id, x, y, z, S.vx[p], S.vy[p], S.vz[p]);
    R.t[k] = c.t; R.ok[k] = c.ok;
    R.contact[3 * k] = c.cx; R.contact[3 * k + 1] = c.cy; R.contact[3 * k + 2] = c.cz;
    R.normal[3 * k] = c.n0; R.normal[3 * k + 1] = c.n1; R.normal[3 * k + 2] = c.n2;
}

// `park` (the gap case with its surface energies still being integrated on the host, amc_wall_park): everything of the
// handler that does not depend on the energy — completed path (old velocity), counters, accumulators zeroed, particle at the
// contact point — and the hit's particle and direction kept in `def_idx` / `def_dir`; k_temp_velocity finishes it.
__global__ __launch_bounds__(256) void k_temp_apply(amc_state S, amc_params P, amc_out O, int case_id, int n,
                                                    temp_records R, const double *__restrict__ dir,
                                                    const double *__restrict__ Es, double *__restrict__ dpz,
                                                    double *__restrict__ dE, int park, int *__restrict__ def_idx,
                                                    double *__restrict__ def_dir)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < 0) n = min(*R.count, R.cap);        // device-RNG mode: the host never saw the count
    if (k >= n) return;
    dpz[k] = 0; dE[k] = 0;
    if (park) { def_idx[k] = R.ok[k] ? R.idx[k] : -1; def_dir[3 * k] = dir[3 * k]; def_dir[3 * k + 1] = dir[3 * k + 1]; def_dir[3 * k + 2] = dir[3 * k + 2]; }
    if (!R.ok[k]) { atomicAdd(&O.banks[amc_bank_id()].n_fp_errors, 1ULL); atomicAdd(&O.banks[amc_bank_id()].n_wall, 1ULL); return; }
    const int p = R.idx[k];
    const double t = R.t[k];
    const double vx = S.vx[p], vy = S.vy[p], vz = S.vz[p];
    double wvx = vx, wvy = vy, wvz = vz;
    const double v_magnitude = park ? sqrt(vx * vx + vy * vy + vz * vz)                       // Temp:377 (as in temp_accommodate)
                                    : temp_accommodate(P, case_id, vx, vy, vz, Es[k], dir[3 * k], dir[3 * k + 1], dir[3 * k + 2],
                                                       wvx, wvy, wvz, dpz[k], dE[k]);
    if (S.flag[p])                                                                           // Temp:391-395
        amc_emit(O, case_id + 1, 0, p, -1, 0, fabs(S.d[p] - fabs(v_magnitude * t)), fabs(S.dx[p] - fabs(vx * t)),
                 fabs(S.dy[p] - fabs(vy * t)), fabs(S.dz[p] - fabs(vz * t)));
    else
        S.flag[p] = 1;
    S.d[p] = 0; S.dx[p] = 0; S.dy[p] = 0; S.dz[p] = 0;                                       // Temp:398-401
    S.x[p] = R.contact[3 * k]; S.y[p] = R.contact[3 * k + 1]; S.z[p] = R.contact[3 * k + 2];   // Temp:402
    S.vx[p] = wvx; S.vy[p] = wvy; S.vz[p] = wvz;                                             // Temp:403
    atomicAdd(&O.banks[amc_bank_id()].n_wall, 1ULL);                                         // Temp:411,482,552
}

// the second half of a parked case: energy accommodation and the new velocity (Temp:377-388) from the velocity the particle
// still has (nothing touched it since it was parked) and the surface energies that have arrived
__global__ __launch_bounds__(256) void k_temp_velocity(amc_state S, amc_params P, int case_id, int n, const int *__restrict__ def_idx,
                                                       const double *__restrict__ def_dir, const double *__restrict__ Es,
                                                       double *__restrict__ dpz, double *__restrict__ dE)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    dpz[k] = 0; dE[k] = 0;
    const int p = def_idx[k];
    if (p < 0) return;                          // (a failed contact solve: counted when the case was parked)
    double wvx, wvy, wvz;
    temp_accommodate(P, case_id, S.vx[p], S.vy[p], S.vz[p], Es[k], def_dir[3 * k], def_dir[3 * k + 1], def_dir[3 * k + 2], wvx, wvy,
                     wvz, dpz[k], dE[k]);
    S.vx[p] = wvx; S.vy[p] = wvy; S.vz[p] = wvz;                                             // Temp:403
}

static temp_records make_records(amc_ctx *c)
{
    temp_records R;
    R.idx = c->T.idx; R.t = c->T.t; R.contact = c->T.contact; R.normal = c->T.normal; R.ok = c->T.ok;
    R.count = c->T.count; R.cap = c->T.cap;
    return R;
}

hipError_t amc_launch_temp_hits(amc_ctx *c, int case_id)
{
    const long long cnt = c->hi - c->lo;
    hipError_t e = hipMemsetAsync(c->T.count, 0, sizeof(int), c->stream);
    if (e != hipSuccess || cnt <= 0) return e;
    AMC_LAUNCH(c, k_temp_hits, dim3((unsigned)((cnt + 255) / 256)), dim3(256), c->S, c->P, case_id,
                       c->lo, c->hi, make_records(c), c->d_cnt);
    return hipGetLastError();
}

hipError_t amc_launch_temp_apply(amc_ctx *c, int case_id, int n, bool park)
{
    if (n <= 0) return hipSuccess;
    AMC_LAUNCH(c, k_temp_apply, dim3((n + 255) / 256), dim3(256), c->S, c->P, c->out, case_id, n,
                       make_records(c), c->T.dir, c->T.Es, c->T.dpz, c->T.dE, park ? 1 : 0, c->T.def_idx, c->T.def_dir);
    return hipGetLastError();
}

hipError_t amc_launch_temp_velocity(amc_ctx *c, int case_id, int n)
{
    if (n <= 0) return hipSuccess;
    AMC_LAUNCH(c, k_temp_velocity, dim3((n + 255) / 256), dim3(256), c->S, c->P, case_id, n, c->T.def_idx, c->T.def_dir,
               c->T.def_Es, c->T.def_dpz, c->T.def_dE);
    return hipGetLastError();
}


// ---- opt-in non-parity mode: directions and energies drawn on the device (include/argonmc.h, amc_temp_rng) -----------
// surface_energy_gap (Temp:143-152): 9 T n k (T/theta)^3 * integral_0^{theta/T} x^3/(e^x - 1) dx, Gauss-Legendre
__device__ inline double temp_gap_energy(const amc_temp_rng &g, double z)
{
    const double m = (g.t_cold - g.t_hot) / g.gap_height;                            // Temp:144
    const double t_gap = m * (z - g.gap_bottom_height) + g.t_hot;                    // Temp:145
    const double X = g.t_debye_alumina / t_gap, half = 0.5 * X;
    double q = 0.0;
    for (int i = 0; i < g.n_gl; i++) {
        const double x = half * (g.gl_x[i] + 1.0);
        q += g.gl_w[i] * (x * x * x / expm1(x));
    }
    q *= half;
    const double r = t_gap / g.t_debye_alumina;
    return 9 * t_gap * g.n_alumina * g.boltzman * (r * r * r) * q;                   // Temp:152
}

// re-emission direction (Temp:119-141 recipe on Philox numbers) and surface energy of one hit
__device__ inline void temp_draw(const amc_params &P, const amc_temp_rng &g, int case_id, unsigned int step, int particle,
                                 double n0, double n1, double n2, double contact_z, double &fx, double &fy, double &fz,
                                 double &Es)
{
    const double cos85 = 0.087155742747658166;      // cos(85 deg), Temp:136
    const double pi = 3.14159265358979323846;
    fx = fy = fz = 0;
    for (unsigned int attempt = 0; attempt < 4096u; attempt++) {                     // Temp:133-141 (acceptance ~91 %)
        unsigned int c[4] = {(unsigned int)particle, step, ((unsigned int)case_id << 16) | attempt, 0x414d4331u};
        philox4x32_10(c, g.seed);
        const double u1 = (double)((((unsigned long long)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0);
        const double u2 = (double)((((unsigned long long)c[2] << 32) | c[3]) >> 12) * (1.0 / 4503599627370496.0);
        const double costheta = -1.0 + 2.0 * u1;                                     // Temp:120  U(-1, 1)
        const double phi = pi * u2;                                                  // Temp:121  U(0, pi)
        const double sgn = (c[3] & 1u) ? 1.0 : -1.0;                                 // Temp:124  choice([-1, 1])
        const double theta = acos(costheta);
        fx = cos(phi) * sin(theta);
        fy = sin(phi) * sin(theta) * sgn;
        fz = cos(theta);
        const double d = fma(fz, n2, fma(fy, n1, fx * n0));
        if (fabs(d) < cos85) continue;                                               // Temp:135-136
        if (d < cos85) { fx = -fx; fy = -fy; fz = -fz; }                             // Temp:138-139
        break;
    }
    Es = (case_id == 5) ? temp_gap_energy(g, contact_z)
                        : ((case_id == 3 || case_id == 7 || case_id == 9) ? P.E_cold : P.E_hot);
}

__global__ __launch_bounds__(256) void k_temp_sample(amc_params P, amc_temp_rng g, int case_id, unsigned int step,
                                                     temp_records R, double *__restrict__ dir, double *__restrict__ Es)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= min(*R.count, R.cap)) return;
    dir[3 * k] = dir[3 * k + 1] = dir[3 * k + 2] = 0.0;
    Es[k] = 0.0;
    if (!R.ok[k]) return;
    double fx, fy, fz, e;
    temp_draw(P, g, case_id, step, R.idx[k], R.normal[3 * k], R.normal[3 * k + 1], R.normal[3 * k + 2], R.contact[3 * k + 2],
              fx, fy, fz, e);
    dir[3 * k] = fx; dir[3 * k + 1] = fy; dir[3 * k + 2] = fz;
    Es[k] = e;
}

// All seven energised cases of a step in ONE pass (device-RNG mode): every case reads and writes only the particle
// itself and the masks are evaluated in case order, each after the previous handler ran (Temp:705-758) — which per
// particle is a sequential evaluation, so with the random numbers available on the device the 7 x (hits, sample,
// apply) kernels collapse into this one.  The per-hit records (one segment per case) are still written: the host
// sums their z-momentum / energy changes in the reference's order, tests read the draws.
struct temp_dev_segments {
    int *idx, *count;
    double *t, *contact, *normal, *dir, *Es, *dpz, *dE;
    unsigned char *ok;
    int cap;
};
__global__ __launch_bounds__(256) void k_temp_all(amc_state S, amc_params P, amc_out O, amc_temp_rng g, unsigned int step,
                                                  long long lo, long long hi, temp_dev_segments D, amc_dev_counters *cnt)
{
    const long long p = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hi) return;
    double x = S.x[p], y = S.y[p], z = S.z[p];
    const double px = S.px[p], py = S.py[p], pz = S.pz[p];
    bool any = false;
    for (int case_id = 3; case_id <= 9; case_id++)
        any |= temp_mask(P, case_id, x, y, z, px, py, pz);
    if (!any) return;                                   // (a hit moves the particle: the masks are re-evaluated below)
    double vx = S.vx[p], vy = S.vy[p], vz = S.vz[p];
    double d = S.d[p], dx = S.dx[p], dy = S.dy[p], dz = S.dz[p];
    bool flag = S.flag[p] != 0;
    int nwall = 0, nerr = 0;
    for (int case_id = 3; case_id <= 9; case_id++) {
        if (!temp_mask(P, case_id, x, y, z, px, py, pz)) continue;
        const int s = case_id - 3;
        const int k = atomicAdd(&D.count[s], 1);
        const bool rec = k < D.cap;
        if (!rec) atomicOr(&cnt->flags, 4ULL);
        const size_t o = (size_t)s * (size_t)D.cap + (size_t)(rec ? k : 0);
        const temp_contact c = temp_solve(P, case_id, x, y, z, vx, vy, vz);
        double fx = 0, fy = 0, fz = 0, es = 0, dpz = 0, dE = 0;
        nwall++;                                                                             // Temp:411,482,552
        if (!c.ok) {
            nerr++;                                                                          // Temp:472-474
        } else {
            temp_draw(P, g, case_id, step, (int)p, c.n0, c.n1, c.n2, c.cz, fx, fy, fz, es);
            double wvx, wvy, wvz;
            const double v_magnitude = temp_accommodate(P, case_id, vx, vy, vz, es, fx, fy, fz, wvx, wvy, wvz, dpz, dE);
            if (flag)                                                                        // Temp:391-395
                amc_emit(O, case_id + 1, 0, (int)p, -1, 0, fabs(d - fabs(v_magnitude * c.t)), fabs(dx - fabs(vx * c.t)),
                         fabs(dy - fabs(vy * c.t)), fabs(dz - fabs(vz * c.t)));
            else
                flag = true;
            d = 0; dx = 0; dy = 0; dz = 0;                                                   // Temp:398-401
            x = c.cx; y = c.cy; z = c.cz;                                                    // Temp:402
            vx = wvx; vy = wvy; vz = wvz;                                                    // Temp:403
        }
        if (rec) {
            D.idx[o] = (int)p; D.t[o] = c.t; D.ok[o] = c.ok;
            D.contact[3 * o] = c.cx; D.contact[3 * o + 1] = c.cy; D.contact[3 * o + 2] = c.cz;
            D.normal[3 * o] = c.n0; D.normal[3 * o + 1] = c.n1; D.normal[3 * o + 2] = c.n2;
            D.dir[3 * o] = fx; D.dir[3 * o + 1] = fy; D.dir[3 * o + 2] = fz;
            D.Es[o] = es; D.dpz[o] = dpz; D.dE[o] = dE;
        }
    }
    S.x[p] = x; S.y[p] = y; S.z[p] = z; S.vx[p] = vx; S.vy[p] = vy; S.vz[p] = vz;
    S.d[p] = d; S.dx[p] = dx; S.dy[p] = dy; S.dz[p] = dz; S.flag[p] = flag ? 1 : 0;
    if (nwall) atomicAdd(&O.banks[amc_bank_id()].n_wall, (unsigned long long)nwall);
    if (nerr) atomicAdd(&O.banks[amc_bank_id()].n_fp_errors, (unsigned long long)nerr);
}

hipError_t amc_launch_temp_cases_device(amc_ctx *c, const amc_temp_rng *cfg)
{
    const long long cnt = c->hi - c->lo;
    amc_temp_dev_ws &D = c->TD;
    hipError_t e = hipMemsetAsync(D.count, 0, sizeof(int) * 7, c->stream);
    if (e != hipSuccess || cnt <= 0) return e;
    static int unfused = -1;
    if (unfused < 0) unfused = getenv("AMC_TEMP_UNFUSED") ? 1 : 0;      // cross-check path: one hits/sample/apply triple per case
    if (!unfused) {
        temp_dev_segments G;
        G.idx = D.idx; G.count = D.count; G.t = D.t; G.contact = D.contact; G.normal = D.normal; G.dir = D.dir;
        G.Es = D.Es; G.dpz = D.dpz; G.dE = D.dE; G.ok = D.ok; G.cap = D.cap;
        AMC_LAUNCH(c, k_temp_all, dim3((unsigned)((cnt + 255) / 256)), dim3(256), c->S, c->P, c->out, *cfg,
                           (unsigned int)c->out.step, c->lo, c->hi, G, c->d_cnt);
        return hipGetLastError();
    }
    const unsigned rec_blocks = (unsigned)((D.cap + 255) / 256);
    for (int s = 0; s < 7; s++) {
        const int case_id = 3 + s;
        const size_t o = (size_t)s * (size_t)D.cap;
        temp_records R;
        R.idx = D.idx + o; R.t = D.t + o; R.contact = D.contact + 3 * o; R.normal = D.normal + 3 * o; R.ok = D.ok + o;
        R.count = D.count + s; R.cap = D.cap;
        AMC_LAUNCH(c, k_temp_hits, dim3((unsigned)((cnt + 255) / 256)), dim3(256), c->S, c->P, case_id,
                           c->lo, c->hi, R, c->d_cnt);
        AMC_LAUNCH(c, k_temp_sample, dim3(rec_blocks), dim3(256), c->P, *cfg, case_id,
                           (unsigned int)c->out.step, R, D.dir + 3 * o, D.Es + o);
        AMC_LAUNCH(c, k_temp_apply, dim3(rec_blocks), dim3(256), c->S, c->P, c->out, case_id, -1, R,
                           D.dir + 3 * o, D.Es + o, D.dpz + o, D.dE + o, 0, (int *)nullptr, (double *)nullptr);
    }
    return hipGetLastError();
}
