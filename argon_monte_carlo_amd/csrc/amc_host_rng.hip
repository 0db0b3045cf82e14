// amc_host_rng.hip — host-only helper of the energised-wall hand-over: the re-emission directions of one case's hits
// (random_components / random_inbounds_direction, Temp:119-141), drawn from the two Mersenne Twisters the reference uses —
// NumPy's legacy global generator and CPython's `random` — whose states the caller passes in NumPy's / CPython's own
// layout (624 key words + position) and gets back advanced.  No GPU code: it is here because the per-hit Python loop
// (three library calls and a small ndarray per attempt, ~4 us) was a third of the energised step's host time.
//
// Bit-for-bit contract, per attempt (checked against the libraries themselves by tests/test_host.py and by the
// self-test in energised.py before the fast path is used at all):
//   costheta = np.random.uniform(-1, 1)  = -1.0 + 2.0 * d,  d = (a * 67108864.0 + b) / 9007199254740992.0,
//              a = next32 >> 5, b = next32 >> 6                 (numpy/random/src/mt19937: mt19937_next_double, random_uniform)
//   phi      = random.uniform(0, pi)     = pi * r,  r = (a * 67108864.0 + b) * (1.0 / 9007199254740992.0) on CPython's twister
//   sign     = np.random.choice([-1, 1]) = legacy randint(0, 2): ONE 32-bit draw & 1 (masked rejection with mask 1)
//   theta = acos(costheta);  F = (cos(phi) sin(theta), sin(phi) sin(theta) sign, cos(theta))  — libm, like math.*
//   s = dot(F, normal): redraw while |s| < cos 85 deg, flip F when s < cos 85 deg
// The dot product is NumPy's (np.dot of two float64[3] = cblas_ddot of its BLAS): the caller hands the function over
// (or selects one of the two plain forms after checking it against np.dot).
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/argonmc.h"

namespace {

struct mt_state { uint32_t *key; int32_t pos; };

inline void mt_regen(uint32_t *mt)
{
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int kk;
    uint32_t y;
    for (kk = 0; kk < 624 - 397; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    for (; kk < 623; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
    }
    y = (mt[623] & UPPER) | (mt[0] & LOWER);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
}

inline uint32_t mt_next(mt_state &s)
{
    if (s.pos >= 624) { mt_regen(s.key); s.pos = 0; }
    uint32_t y = s.key[s.pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

typedef double (*ddot64_fn)(int64_t, const double *, int64_t, const double *, int64_t);
typedef double (*ddot32_fn)(int, const double *, int, const double *, int);

}  // namespace

extern "C" int amc_host_directions(uint32_t *np_key, int32_t *np_pos, uint32_t *py_key, int32_t *py_pos,
                                   const double *normal_xyz, const uint8_t *ok, int64_t n, double cos85, double pi,
                                   int dot_kind, void *dot_fn, double *dir_xyz)
{
    if (!np_key || !np_pos || !py_key || !py_pos || (n > 0 && (!normal_xyz || !dir_xyz))) return AMC_ERR_INVALID;
    if (*np_pos < 0 || *np_pos > 624 || *py_pos < 0 || *py_pos > 624) return AMC_ERR_INVALID;
    if ((dot_kind == 2 || dot_kind == 3) && !dot_fn) return AMC_ERR_INVALID;
    mt_state NP = {np_key, *np_pos}, PY = {py_key, *py_pos};
    for (int64_t k = 0; k < n; k++) {
        double *out = dir_xyz + 3 * k;
        if (ok && !ok[k]) { out[0] = out[1] = out[2] = 0.0; continue; }      // (the reference fails before any draw there)
        const double *nm = normal_xyz + 3 * k;
        for (;;) {
            const double a1 = (double)(int32_t)(mt_next(NP) >> 5), b1 = (double)(int32_t)(mt_next(NP) >> 6);
            const double costheta = -1.0 + 2.0 * ((a1 * 67108864.0 + b1) / 9007199254740992.0);
            const double a2 = (double)(mt_next(PY) >> 5), b2 = (double)(mt_next(PY) >> 6);
            const double phi = pi * ((a2 * 67108864.0 + b2) * (1.0 / 9007199254740992.0));
            const double theta = acos(costheta);
            const double sign = (mt_next(NP) & 1u) ? 1.0 : -1.0;
            const double st = sin(theta);
            double F[3];
            F[0] = cos(phi) * st;
            F[1] = (sin(phi) * st) * sign;
            F[2] = cos(theta);
            double s;
            switch (dot_kind) {
            case 2: s = ((ddot64_fn)dot_fn)(3, F, 1, nm, 1); break;
            case 3: s = ((ddot32_fn)dot_fn)(3, F, 1, nm, 1); break;
            case 1: s = fma(F[2], nm[2], fma(F[1], nm[1], F[0] * nm[0])); break;
            default: s = F[0] * nm[0]; s = s + F[1] * nm[1]; s = s + F[2] * nm[2]; break;
            }
            if (fabs(s) < cos85) continue;
            if (s < cos85) { F[0] = -F[0]; F[1] = -F[1]; F[2] = -F[2]; }
            out[0] = F[0]; out[1] = F[1]; out[2] = F[2];
            break;
        }
    }
    *np_pos = NP.pos; *py_pos = PY.pos;
    return AMC_OK;
}
