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

__global__ __launch_bounds__(256) void k_temp_hits(amc_state S, amc_params P, int case_id, long long lo, long long hi,
                                                   temp_records R, amc_dev_counters *cnt)
{
    const long long p = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hi) return;
    const double x = S.x[p], y = S.y[p], z = S.z[p];
    if (!temp_mask(P, case_id, x, y, z, S.px[p], S.py[p], S.pz[p])) return;
    const int k = atomicAdd(R.count, 1);
    if (k >= R.cap) { atomicOr(&cnt->flags, 4ULL); return; }
    const double vx = S.vx[p], vy = S.vy[p], vz = S.vz[p];
    R.idx[k] = (int)p;
    unsigned char ok = 1;
    double t = 0, cx = 0, cy = 0, cz = 0, n0 = 0, n1 = 0, n2 = 0;
    if (case_id == 3 || case_id == 4 || case_id == 6 || case_id == 7) {
        const double zp = case_id == 3 ? P.t_z3_cold : case_id == 4 ? P.t_z3_hot : case_id == 6 ? P.t_zgap_lo : P.t_zgap_hi;
        t = (z - zp) / vz;                                                                   // Temp:353
        cx = x - vx * t; cy = y - vy * t; cz = zp;                                           // Temp:372
        n2 = (case_id == 3 || case_id == 6) ? 1.0 : -1.0;                                    // Temp:709,714,730,736
    } else {
        const double Rc = case_id == 5 ? P.R_g_c : P.R_p_c;
        const double a = (-vx) * (-vx) + (-vy) * (-vy);                                      // Temp:436
        const double b = 2 * (x * (-vx) + y * (-vy));
        const double c = x * x + y * y - Rc * Rc;
        const double disc2 = b * b - 4 * a * c;
        if (a == 0.0 || disc2 < 0.0 || disc2 != disc2) {
            ok = 0;                                                                          // Temp:472-474
        } else {
            const double sq = sqrt(disc2);
            const double t1 = (-b + sq) / (2 * a), t2 = (-b - sq) / (2 * a);
            t = (t1 < t2) ? t1 : t2;                                                         // Temp:439
            cx = x - vx * t; cy = y - vy * t; cz = z - vz * t;                               // Temp:440
            n0 = -(cx / Rc); n1 = -(cy / Rc); n2 = -(0.0 / Rc);                              // Temp:442-444 (negated)
        }
    }
    R.t[k] = t; R.ok[k] = ok;
    R.contact[3 * k] = cx; R.contact[3 * k + 1] = cy; R.contact[3 * k + 2] = cz;
    R.normal[3 * k] = n0; R.normal[3 * k + 1] = n1; R.normal[3 * k + 2] = n2;
}

__global__ __launch_bounds__(256) void k_temp_apply(amc_state S, amc_params P, amc_out O, int case_id, int n,
                                                    temp_records R, const double *__restrict__ dir,
                                                    const double *__restrict__ Es, double *__restrict__ dpz,
                                                    double *__restrict__ dE)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    dpz[k] = 0; dE[k] = 0;
    if (!R.ok[k]) { atomicAdd(&O.cnt->n_fp_errors, 1ULL); atomicAdd(&O.cnt->n_wall, 1ULL); return; }
    const int p = R.idx[k];
    const double m = P.argon_mass;
    const double alpha = (case_id == 5) ? P.alpha_gap : P.alpha_coated;
    const double t = R.t[k];
    const double vx = S.vx[p], vy = S.vy[p], vz = S.vz[p];
    const double v_magnitude = sqrt(vx * vx + vy * vy + vz * vz);                            // Temp:377
    const double old_pz = m * vz;                                                            // Temp:378
    const double E = 0.5 * m * (v_magnitude * v_magnitude);                                  // Temp:128-129,379
    const double diff = Es[k] - E;                                                           // Temp:380
    const double Enew = E + diff * alpha;                                                    // Temp:381
    const double mag = sqrt(Enew * 2 / m);                                                   // Temp:383
    dE[k] = Enew - E;                                                                        // Temp:384
    const double wvx = dir[3 * k] * mag, wvy = dir[3 * k + 1] * mag, wvz = dir[3 * k + 2] * mag;   // Temp:386
    dpz[k] = m * wvz - old_pz;                                                               // Temp:387-388
    if (S.flag[p])                                                                           // Temp:391-395
        amc_emit(O, case_id + 1, 0, p, -1, 0, fabs(S.d[p] - fabs(v_magnitude * t)), fabs(S.dx[p] - fabs(vx * t)),
                 fabs(S.dy[p] - fabs(vy * t)), fabs(S.dz[p] - fabs(vz * t)));
    else
        S.flag[p] = 1;
    S.d[p] = 0; S.dx[p] = 0; S.dy[p] = 0; S.dz[p] = 0;                                       // Temp:398-401
    S.x[p] = R.contact[3 * k]; S.y[p] = R.contact[3 * k + 1]; S.z[p] = R.contact[3 * k + 2];   // Temp:402
    S.vx[p] = wvx; S.vy[p] = wvy; S.vz[p] = wvz;                                             // Temp:403
    atomicAdd(&O.cnt->n_wall, 1ULL);                                                         // Temp:411,482,552
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
    hipLaunchKernelGGL(k_temp_hits, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->S, c->P, case_id,
                       c->lo, c->hi, make_records(c), c->d_cnt);
    return hipGetLastError();
}

hipError_t amc_launch_temp_apply(amc_ctx *c, int case_id, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_temp_apply, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->S, c->P, c->out, case_id, n,
                       make_records(c), c->T.dir, c->T.Es, c->T.dpz, c->T.dE);
    return hipGetLastError();
}
