// amc_ic.hip — synthetic initial conditions generated on the device (include/argonmc.h, amc_ic_config).
// The recipe of the reference's generators (Cube:144-172, Pore:106-158) as restated in argon_monte_carlo_amd/ic.py —
// positions uniform per region (cube, or stacked cylinders with r = R sqrt(u), theta ~ U(0, 2 pi), z ~ U(z_lo, z_hi),
// Pore:120-139), velocity components N(0, a_shape^2) (= Maxwell speeds, isotropic; Pore:144-158), accumulators zero, flag
// clear — on a counter-based generator of its own: a particle's numbers depend on (seed, particle index) only, so every
// rank of a sharded run produces the identical system without moving 137 B per particle through PCIe.  NOT the
// reference's scipy / NumPy streams: opt-in, never what a parity test starts from.
#include "amc_host.h"
#include "amc_philox.h"

#define AMC_IC_TAG 0x414d4349u      // "AMCI": keeps these counters apart from the energised-wall draws

// two uniform doubles in [0, 1) with 53 bits each from Philox block `blk` of particle p
__device__ inline void ic_uniform2(unsigned long long seed, unsigned int p, unsigned int blk, double &u, double &v)
{
    unsigned int c[4] = {p, blk, 0u, AMC_IC_TAG};
    philox4x32_10(c, seed);
    u = (double)((((unsigned long long)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0);
    v = (double)((((unsigned long long)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0);
}

__global__ __launch_bounds__(256) void k_ic(amc_state S, amc_params P, amc_ic_config C, long long n)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double two_pi = 6.283185307179586;
    double u0, u1, u2, u3, u4, u5, u6, u7;
    ic_uniform2(C.seed, (unsigned int)p, 0u, u0, u1);
    ic_uniform2(C.seed, (unsigned int)p, 1u, u2, u3);
    ic_uniform2(C.seed, (unsigned int)p, 2u, u4, u5);
    ic_uniform2(C.seed, (unsigned int)p, 3u, u6, u7);
    double x, y, z;
    if (C.n_regions == 0) {
        x = u0 * P.cube_x; y = u1 * P.cube_y; z = u2 * P.cube_z;
    } else {
        int r = 0;
        while (r + 1 < C.n_regions && p >= C.first[r + 1]) r++;
        const double th = two_pi * u0, rad = C.radius[r] * sqrt(u1);
        x = rad * cos(th); y = rad * sin(th);
        z = C.z_lo[r] + (C.z_hi[r] - C.z_lo[r]) * u2;
    }
    // Box-Muller on (1 - u) in (0, 1]
    const double m0 = sqrt(-2.0 * log(1.0 - u3)), m1 = sqrt(-2.0 * log(1.0 - u5));
    S.x[p] = x; S.y[p] = y; S.z[p] = z;
    S.vx[p] = C.a_shape * m0 * cos(two_pi * u4);
    S.vy[p] = C.a_shape * m0 * sin(two_pi * u4);
    S.vz[p] = C.a_shape * m1 * cos(two_pi * u6);
    S.d[p] = 0.0; S.dx[p] = 0.0; S.dy[p] = 0.0; S.dz[p] = 0.0;
    S.flag[p] = 0;
}

extern "C" int amc_init_synthetic(amc_ctx *c, const amc_ic_config *cfg)
{
    if (!c || !cfg || cfg->struct_size != (int32_t)sizeof(amc_ic_config)) return AMC_ERR_INVALID;
    if (cfg->n_regions < 0 || cfg->n_regions > 8 || !(cfg->a_shape >= 0.0)) return amc_fail(c, AMC_ERR_INVALID, "amc_init_synthetic: bad configuration");
    if (c->n > 0xffffffffLL) return amc_fail(c, AMC_ERR_INVALID, "amc_init_synthetic: particle index exceeds the 32-bit counter word");
    c->lists_age = -1;          // (kept lists: a new state starts with a full build)
    if (cfg->n_regions == 0 && c->P.geometry != AMC_GEOM_CUBE && c->P.geometry != AMC_GEOM_CELL)
        return amc_fail(c, AMC_ERR_INVALID, "amc_init_synthetic: this geometry needs the region table");
    for (int r = 0; r < cfg->n_regions; r++)
        if (cfg->first[r] > cfg->first[r + 1] || cfg->first[0] != 0 || !(cfg->radius[r] >= 0.0) || !(cfg->z_hi[r] >= cfg->z_lo[r]))
            return amc_fail(c, AMC_ERR_INVALID, "amc_init_synthetic: region %d is malformed", r);
    if (cfg->n_regions > 0 && cfg->first[cfg->n_regions] != c->n) return amc_fail(c, AMC_ERR_INVALID, "amc_init_synthetic: the regions hold %lld particles, the context %lld", (long long)cfg->first[cfg->n_regions], (long long)c->n);
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    if (c->n > 0) {
        AMC_LAUNCH(c, k_ic, dim3((unsigned)((c->n + 255) / 256)), dim3(256), c->S, c->P, *cfg, (long long)c->n);
        AMC_HIP(c, hipGetLastError());
    }
    c->uploaded = true;
    return amc_publish_velocities(c);
}
