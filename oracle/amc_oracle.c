/*
 * amc_oracle.c — CPU restatement of the reference's hot path (drift -> walls -> bounds check -> p-p sweep),
 * plain C, single thread.
 *
 *   THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *   Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load liboracle.so, and only as
 *   the checker / the reported CPU baseline.  libargonmc.so (the product) never links, loads or calls it and has
 *   no CPU fallback.
 *
 * Parity status: PINNED.  orc_pow_* is checked bit-for-bit (positions, velocities, path accumulators, flags,
 * completed-path lists, collision counters) against outputs of the reference itself, produced in the build
 * container by oracle/gen_golden.py (imports /root/reference/Open_Air_Pore_MC.py for the function-level vectors
 * and runs patched temporary copies of the three scripts for the step-level dumps) and committed under
 * tests/golden/.  See tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it follows (Cube = Open_Air_Cube_MC.py, Pore = Open_Air_Pore_MC.py,
 * Temp = Temperature_Pore_MC.py).  No reference source text is copied; the arithmetic order is restated from
 * SURVEY.md App. A/B and from reading the scripts.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/argonmc.h"

/* state = the reference's module-global arrays (Pore:385-400) */
typedef struct orc_state {
    int64_t n;
    double *x, *y, *z, *vx, *vy, *vz;      /* x_vals.., x_velocities..                        */
    double *d, *dx, *dy, *dz;              /* dist_since_collision, dist_{x,y,z}_since_collision */
    uint8_t *flag;                         /* full_path_traveled                              */
    double *px, *py, *pz;                  /* prior_{x,y,z}_vals (pore geometries)            */
} orc_state;

/* completed_paths / completed_{x,y,z}_paths (Pore:408-413), appended in the reference's own order */
typedef struct orc_sink {
    amc_path_record *rec;
    int64_t cap, n;
    int64_t overflow;
} orc_sink;

static void orc_emit(orc_sink *s, int32_t step, int32_t phase, int64_t cell, int32_t i, int32_t j, int32_t which,
                     double tot, double px, double py, double pz)
{
    if (!s) return;
    if (s->n >= s->cap) { s->overflow++; return; }
    amc_path_record *r = &s->rec[s->n++];
    r->step = step; r->phase = phase; r->cell = cell; r->i = i; r->j = j; r->which = which; r->reserved = 0;
    r->total = tot; r->px = px; r->py = py; r->pz = pz;
}

#define ORC(name) orc_pow_##name
#define SQ(x) pow((x), 2.0)
#include "amc_oracle_impl.h"
#undef ORC
#undef SQ

#define ORC(name) orc_mul_##name
#define SQ(x) ((x) * (x))
#include "amc_oracle_impl.h"
#undef ORC
#undef SQ

#ifdef _OPENMP
#include <omp.h>
#endif
/* threads of the *_par functions (libgomp reads OMP_NUM_THREADS when it is loaded: too early for a caller that finds
 * out its CPU share at run time) */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_abi_version(void) { return AMC_ABI_VERSION; }
int64_t orc_sizeof_params(void) { return (int64_t)sizeof(amc_params); }
int64_t orc_sizeof_path_record(void) { return (int64_t)sizeof(amc_path_record); }
int64_t orc_sizeof_step_stats(void) { return (int64_t)sizeof(amc_step_stats); }
