/*
 * argonmc.h — C ABI of libargonmc.so: the MI355X-native (gfx950, HIP) drift -> wall-reflection ->
 * particle-particle collision sweep of the hard-sphere argon Monte Carlo.
 *
 * The reference (Lightbrite88/Argon_Monte_Carlo) is pure Python and has no FFI.  Every entry point below
 * therefore cites the reference code it REPLACES (file:line under /root/reference; Cube = Open_Air_Cube_MC.py,
 * Pore = Open_Air_Pore_MC.py, Temp = Temperature_Pore_MC.py); INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (AMC_OK) or a negative amc_status; nothing throws across the boundary;
 *     amc_last_error(ctx) returns a string owned by the ctx (valid until the next call on that ctx).
 *   - a ctx is driven by ONE host thread.  Host buffers are C-contiguous float64 / uint8 / int32 arrays owned by
 *     the caller; the library copies in/out and never retains host pointers.
 *   - all arithmetic is IEEE double, compiled with -ffp-contract=off; see DESIGN.md "numerics".
 *   - there is NO CPU fallback: if no HIP device is usable, amc_create fails with AMC_ERR_NO_DEVICE.
 */
#ifndef ARGONMC_H
#define ARGONMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMC_ABI_VERSION 2

typedef enum amc_status {
    AMC_OK = 0,
    AMC_ERR_INVALID = -1,      /* bad argument / params                                            */
    AMC_ERR_NO_DEVICE = -2,    /* no usable HIP device (the product path never falls back to CPU)   */
    AMC_ERR_HIP = -3,          /* a HIP runtime call failed; see amc_last_error                      */
    AMC_ERR_CAPACITY = -4,     /* a device work buffer overflowed (candidates, clusters, paths)       */
    AMC_ERR_FP = -5,           /* the reference would have raised FloatingPointError (np.seterr(all='raise'),
                                  Pore:11 / Temp:15): zero relative velocity in the contact solve etc. */
    AMC_ERR_STATE = -6         /* call made in the wrong state (e.g. timestep before upload)          */
} amc_status;

/* Geometry kinds = which reference loop body one amc_timestep() reproduces. */
typedef enum amc_geometry {
    AMC_GEOM_CELL = 0,            /* a single cell: pairwise_particles_in_cell only (Pore:160-255)       */
    AMC_GEOM_CUBE = 1,            /* Open_Air_Cube_MC.py loop body, Cube:175-338                         */
    AMC_GEOM_PORE = 2,            /* Open_Air_Pore_MC.py loop body, Pore:416-557 (specular walls)        */
    AMC_GEOM_PORE_ENERGISED = 3   /* Temperature_Pore_MC.py loop body, Temp:662-853                      */
} amc_geometry;

/*
 * Constants of one simulation.  The HOST evaluates every threshold with exactly the expression the reference
 * writes (e.g. Pore's gap top is (total_height - open_air_height) - cold_coating_height = 1.6000000000000059e-07,
 * Temp's is open_air_height + hot_coating_height + gap_height = 1.6000000000000003e-07) and passes the resulting
 * doubles; device code never re-derives them.  argon_monte_carlo_amd/params.py builds this struct.
 */
typedef struct amc_params {
    int32_t struct_size;          /* = sizeof(amc_params), ABI check                                   */
    int32_t geometry;             /* amc_geometry                                                      */
    int64_t n;                    /* number of particles (num_molecules, Pore:64)                      */

    double collision_range;       /* Pore:58                                                           */
    double argon_mass;            /* Pore:49                                                           */
    double argon_radius;          /* Pore:56                                                           */

    /* reference cell grid of the p-p sweep (Cube:30-38,233-237; Pore:41-46,527-529) */
    int32_t nx, ny, nz;           /* num_{x,y,z}_subdivions                                            */
    int32_t reserved0;            /* flags: bit0 = keep prior_{x,y,z}_vals (amc_download_prior)            */
    double dx, dy, dz;
    double overlap_x, overlap_y, overlap_z;   /* Cube: d/10 ; Pore/Temp: collision_range                */

    /* Cube walls (Cube:189-226) */
    double cube_x, cube_y, cube_z;

    /* Pore / Temp radii and heights (Pore:25-39,67-69) */
    double R_oa, R_oa_c;          /* open_air_radius, open_air_collision_radius                        */
    double R_p, R_p_c;            /* pore_coated_radius, pore_collision_radius                         */
    double R_g, R_g_c;            /* gap_radius, gap_collision_radius                                  */
    double H;                     /* total_height                                                      */
    double h_oa;                  /* open_air_height                                                   */
    double z_cold;                /* total_height - open_air_height                                    */
    double z_gap_bottom;          /* open_air_height + hot_coating_height                              */
    double z_gap_top;             /* Pore: total_height-open_air_height-cold_coating_height ; Temp: gap_top_height */

    /* bounds check / recapture (Pore:354-375; Temp:594-616) */
    double oob_z_lo_fix;          /* Pore: 10*argon_radius (added)      ; Temp: 50e-9 (assigned)       */
    double oob_z_hi_fix;          /* Pore: 10*argon_radius (subtracted) ; Temp: total_height-50e-9      */
    double R_oa_sq, R_g_sq, R_p_sq;           /* open_air_radius**2 etc. as Python evaluates them      */
    double z_oob_hot_top;         /* open_air_height + hot_coating_height                              */
    double z_oob_gap_top;         /* open_air_height + hot_coating_height + gap_height                 */

    /* Temp-only thresholds (Temp:690-753), all pre-evaluated */
    double t_z3_cold;             /* total_height - open_air_height + argon_radius                     */
    double t_z3_hot;              /* open_air_height - argon_radius                                    */
    double t_zgap_lo;             /* gap_bottom_height + argon_radius                                  */
    double t_zgap_hi;             /* gap_top_height - argon_radius                                     */
    double R_g_c_sq, R_p_c_sq;    /* gap_collision_radius**2, pore_collision_radius**2                 */
    double E_cold, E_hot;         /* surface_energy_{cold,hot} (Temp:83-84)                            */
    double alpha_coated, alpha_gap;           /* Temp:76-77                                            */
    double cos85;                 /* cos(85*pi/180) (Temp:136)                                         */

    /* free-path histogram (Pore:93,575): nbins equal bins on [hist_lo, hist_hi] */
    int32_t hist_bins;
    int32_t reserved1;            /* flags: bit0 = count-and-continue on a failed wall/contact solve (Temp:340-342)
                                     instead of reporting AMC_ERR_FP (what Pore:336-338 / np.seterr amount to)   */
    double hist_lo, hist_hi;

    /* engine knobs (not physics) */
    double fine_cell;             /* edge of the detection grid cells (m); 0 = choose automatically     */
    int32_t device;               /* HIP device ordinal                                                */
    int32_t detect_mode;          /* 0 auto, 1 binned (cell grid), 2 tiled all-pairs detector (no grid at
                                   * all up to 4096 particles; above, in front of the grid-based resolve) */
    int64_t max_candidates;       /* capacity of the candidate-pair list; 0 = default                  */
    int64_t max_paths;            /* capacity of the completed-path record buffer; 0 = default (2^20), < 0 = no
                                   * records at all (completed paths only go into the device histograms)     */
} amc_params;

/* Per-step counters.  n_collisions = what the reference accumulates in num_collisions_per_step (Pore:424,
 * 244-245, 292, 348, 556): wall hits that count + p-p collisions. */
typedef struct amc_step_stats {
    int64_t n_pp;                 /* particle-particle collisions (Pore:241)                           */
    int64_t n_wall;               /* wall hits counted by the reference (Pore:292,348; Temp:411,482,552) */
    int64_t n_oob_walls;          /* particles moved by the bounds check after the wall stage (Pore:512; Temp:804) */
    int64_t n_oob_pp;             /* ... after the p-p sweep (Pore:550; Temp:844)                      */
    int64_t n_paths;              /* completed free paths emitted this step                            */
    int64_t n_candidates;         /* close pairs found by the detect kernel                            */
    int64_t n_clusters;           /* interaction clusters resolved                                     */
    int64_t n_rounds;             /* resolve validation rounds (1 = no cluster merge was needed)       */
    int64_t n_fp_errors;          /* events where the reference would raise (a == 0, negative discriminant) */
    int64_t flags;                /* bit0 candidate overflow, bit1 path overflow, bit2 cluster overflow, bit4 velocity-change list (multi-GPU) */
} amc_step_stats;

/* One completed free path (Pore:186-199, 274-278, 324-328).  (step, phase, cell, i, j, which) is the position of
 * the append in the reference's own order, so sorting records by those keys reproduces its list order:
 *   phase 1..9  = wall cases in evaluation order (Pore:442-485 / Temp:693-753), i = particle, j = -1
 *   phase 16+g  = p-p colour group g = 4*x_group+2*y_group+z_group (Pore:522-524); Cube uses phase 16
 *   cell        = linear reference-cell index ((lx*ny)+ly)*nz+lz ; which = 0 for particle j, 1 for particle i */
typedef struct amc_path_record {
    int32_t step;
    int32_t phase;
    int64_t cell;
    int32_t i;
    int32_t j;
    int32_t which;
    int32_t reserved;
    double total, px, py, pz;
} amc_path_record;

typedef struct amc_ctx amc_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
int amc_abi_version(void);
int amc_create(amc_ctx **out, const amc_params *p);
void amc_destroy(amc_ctx *ctx);
const char *amc_last_error(const amc_ctx *ctx);
/* Run all subsequent work on this hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream. */
int amc_set_stream(amc_ctx *ctx, void *hip_stream);
/* Run on HIP's NULL (legacy default) stream — what torch.cuda.current_stream() is unless the caller changed it; needed
 * so that torch.distributed collectives and this library's kernels are ordered on one stream. */
int amc_use_null_stream(amc_ctx *ctx);
int amc_synchronize(amc_ctx *ctx);

/* ---- state (replaces the module-global ndarrays of Pore:385-400) ----------------------------------------- */
/* x_vals,y_vals,z_vals,x_velocities,y_velocities,z_velocities, dist_since_collision, dist_{x,y,z}_since_collision,
 * full_path_traveled — float64[n] each, uint8[n] for the flag.  Any pointer may be NULL = leave unchanged / skip. */
int amc_upload(amc_ctx *ctx, const double *x, const double *y, const double *z, const double *vx, const double *vy,
               const double *vz, const double *dist, const double *dist_x, const double *dist_y,
               const double *dist_z, const uint8_t *full_path);
int amc_download(amc_ctx *ctx, double *x, double *y, double *z, double *vx, double *vy, double *vz, double *dist,
                 double *dist_x, double *dist_y, double *dist_z, uint8_t *full_path);
/* prior_{x,y,z}_vals (Pore:427-429) of the last step */
int amc_download_prior(amc_ctx *ctx, double *px, double *py, double *pz);

/* ---- the hot path -------------------------------------------------------------------------------------- */
/* One iteration of the reference's time loop body: drift (Pore:426-437) -> wall cases (Pore:439-485 | Cube:189-226)
 * -> bounds check (Pore:512) -> p-p sweep (Pore:520-549 | Cube:231-336) -> bounds check (Pore:550).
 * Blocks until the step is done and returns its counters. */
int amc_timestep(amc_ctx *ctx, double dt, amc_step_stats *out);
/* nsteps iterations enqueued back-to-back with no host synchronisation in between; *sum receives the counters
 * summed over the steps (may be NULL).  Not available for AMC_GEOM_PORE_ENERGISED (host RNG handshake per step). */
int amc_run(amc_ctx *ctx, double dt, int64_t nsteps, amc_step_stats *sum);

/* individual stages, for function-level parity tests against the reference's handlers */
int amc_stage_drift(amc_ctx *ctx, double dt);                 /* Pore:426-437 / Cube:179-187            */
int amc_stage_walls(amc_ctx *ctx, amc_step_stats *out);      /* Pore:439-485 / Cube:189-226             */
int amc_stage_bounds(amc_ctx *ctx, int64_t *n_moved);        /* Pore:354-375 / Temp:594-616             */
int amc_stage_sweep(amc_ctx *ctx, amc_step_stats *out);      /* Pore:520-549 / Cube:231-336             */

/* Direct replacement of pairwise_particles_in_cell (Pore:160-255 == Temp:215-309 == Cube:253-324) for ONE cell:
 * arrays of length n_cell are updated in place exactly as the reference returns them; completed paths are
 * appended in the reference's order to out_paths[4][cap] (total,x,y,z rows); returns the collision count that the
 * reference adds to num_collisions_per_step.  Needs a ctx only for its device/stream and scratch. */
int amc_pairwise_cell(amc_ctx *ctx, int64_t n_cell, double *continue_path, double *continue_x_path,
                      double *continue_y_path, double *continue_z_path, uint8_t *has_collided, double *x, double *y,
                      double *z, double *vx, double *vy, double *vz, double *out_paths, size_t cap,
                      size_t *n_paths, int64_t *n_collisions);

/* ---- energised walls: host-RNG handshake (Temp:132-141 consumes two Mersenne-Twister streams in particle order,
 * Temp:147-152 calls mpmath.quad per gap hit — both stay on the host; argon_monte_carlo_amd/energised.py) -----------
 * One step of Temperature_Pore_MC.py (Temp:662-853) is
 *     amc_temp_begin(dt);  for case in 3..9: amc_wall_hits(case) -> host draws -> amc_wall_apply(case);  amc_temp_end()
 * case ids in the reference's evaluation order (Temp:708-751): 3 case-3 cold plane, 4 case-3 hot plane, 5 case-4 gap
 * side wall, 6 case-5 bottom plane, 7 case-5 top plane, 8 case-6 hot side wall, 9 case-6 cold side wall. */
/* drift (Temp:672-683) + specular cases 1-2 (Temp:693-703) */
int amc_temp_begin(amc_ctx *ctx, double dt);
/* Evaluates the mask of energised case `case_id` on the current state; returns the hit particles in ASCENDING index
 * with the inward unit normal handed to random_inbounds_direction (Temp:375,444; (0,0,0) marks a hit whose contact
 * solve fails, Temp:472-474: no draw is consumed for it) and the contact z (surface_energy_gap argument, Temp:519). */
int amc_wall_hits(amc_ctx *ctx, int case_id, int32_t *idx, double *normal_xyz, double *contact_z, size_t cap,
                  size_t *n);
/* Re-emission of those hits: unit direction (3 per hit) and surface energy per hit, in the same order; writes the
 * per-hit z-momentum and energy changes (Temp:384-389), which the caller sums left to right like the reference. */
int amc_wall_apply(amc_ctx *ctx, int case_id, const double *dir_xyz, const double *surface_energy, size_t n,
                   double *dpz, double *dE);
/* A case whose surface energies are still being computed (the gap case: mpmath.quad per hit, Temp:143-152, in worker
 * processes) can be PARKED: amc_wall_park does everything of amc_wall_apply that does not depend on the energy (completed
 * path, counters, particle at its contact point — what the following cases' masks read), amc_wall_finish sets the new
 * velocities and returns the per-hit changes once the energies are there.  Exact as long as no later case of the step hits
 * a parked particle in between: the caller compares the hit lists, finishes first if one does and calls
 * amc_wall_hits_again so that the next amc_wall_hits evaluates its case anew on the finished state. */
int amc_wall_park(amc_ctx *ctx, int case_id, const double *dir_xyz, size_t n);
int amc_wall_finish(amc_ctx *ctx, int case_id, const double *surface_energy, size_t n, double *dpz, double *dE);
int amc_wall_hits_again(amc_ctx *ctx);
/* recapture (Temp:804) -> p-p sweep (Temp:813-842) -> recapture (Temp:844); returns the step's counters */
int amc_temp_end(amc_ctx *ctx, amc_step_stats *out);

/* ---- outputs -------------------------------------------------------------------------------------------- */
/* completed_paths / completed_{x,y,z}_paths (Pore:408-413) since the last drain, unsorted */
int amc_drain_paths(amc_ctx *ctx, amc_path_record *out, size_t cap, size_t *n);
int amc_paths_pending(amc_ctx *ctx, size_t *n);
/* free-path histograms accumulated on the device with np.histogram(range=(lo,hi), bins) semantics
 * (Pore:575-596): counts[4][hist_bins] (total, x, y, z) and the number of paths seen (incl. out of range). */
int amc_histograms(amc_ctx *ctx, uint64_t *counts, uint64_t *n_paths_total);
int amc_reset_outputs(amc_ctx *ctx);

/* Host-only helper of the hand-over above (no GPU, no context): the re-emission directions of one case's hits —
 * random_components / random_inbounds_direction (Temp:119-141) — drawn from the reference's two Mersenne Twisters, whose
 * states come in NumPy's / CPython's layout (uint32 key[624] + position 0..624: np.random.get_state()[1:3],
 * random.getstate()[1]) and go back advanced exactly as the per-hit library calls would have left them.
 * normal_xyz [n][3]; ok == NULL or ok[k] == 0: no draw, zero direction (the reference fails before it draws, Temp:367);
 * cos85 = cos(85 pi / 180), pi = math.pi as the caller's Python computes them; dir_xyz [n][3] out.
 * dot_kind selects how dot(direction, normal) is formed — it must equal np.dot of two float64[3] on the caller's NumPy:
 *   0: ((x0 y0) + x1 y1) + x2 y2   1: fma(x2, y2, fma(x1, y1, x0 y0))
 *   2: dot_fn = cblas_ddot with 64-bit integers   3: dot_fn = cblas_ddot with 32-bit integers
 * (argon_monte_carlo_amd/energised.py takes the BLAS NumPy itself loaded and checks the whole function against the
 * per-hit library calls before it uses it). */
int amc_host_directions(uint32_t *np_key, int32_t *np_pos, uint32_t *py_key, int32_t *py_pos, const double *normal_xyz,
                        const uint8_t *ok, int64_t n, double cos85, double pi, int dot_kind, void *dot_fn, double *dir_xyz);

/* ---- opt-in, NON-PARITY mode: the energised cases entirely on the device ----------------------------------
 * The reference draws the re-emission direction of every hit from two Mersenne Twisters in particle order with a
 * rejection loop (Temp:119-141) and integrates the gap wall's surface energy with mpmath (Temp:143-152); reproducing
 * its numbers needs the host hand-over above.  This mode keeps the recipe and replaces the generators: a counter-based
 * Philox4x32-10 keyed by `seed`, counter = (particle, step, case, attempt) — results do not depend on hit order, shard
 * layout or launch geometry — and a Gauss-Legendre rule for the Debye integral.  Everything else (masks, contact
 * points, accommodation, bookkeeping) is the same code as the parity path.
 *   amc_temp_begin(dt) -> amc_temp_cases_device(cfg) -> amc_temp_end(&stats), no host synchronisation in between;
 *   amc_temp_device_results(case, ...) afterwards returns the per-hit z-momentum / energy changes of that case in
 *   ascending particle index (the order Temp:385-389 sums them in); amc_temp_device_draws is for inspection / tests. */
typedef struct amc_temp_rng {
    int32_t struct_size;          /* sizeof(amc_temp_rng)                                                        */
    int32_t n_gl;                 /* Gauss-Legendre points in use, 2..32                                         */
    uint64_t seed;
    double t_cold, t_hot, gap_height, gap_bottom_height;      /* Temp:31-32, 44-45, 144-145                      */
    double t_debye_alumina, n_alumina, boltzman;              /* Temp:148-152                                    */
    double gl_x[32], gl_w[32];    /* nodes and weights on [-1, 1]                                                */
} amc_temp_rng;
int amc_temp_cases_device(amc_ctx *ctx, const amc_temp_rng *cfg);
int amc_temp_device_results(amc_ctx *ctx, int case_id, int32_t *idx, double *dpz, double *dE, uint8_t *ok, size_t cap, size_t *n);
/* the step's sums of those per-hit changes, formed like Temp:385-389 / 705-758 do (left to right in ascending particle
 * index inside a case, cases in order): sums[0] = z-momentum, [1] = energy to the cold walls, [2] = to the hot walls;
 * had[k] != 0 if any hit contributed to sums[k] */
int amc_temp_device_sums(amc_ctx *ctx, double *sums /*[3]*/, int32_t *had /*[3]*/);
int amc_temp_device_draws(amc_ctx *ctx, int case_id, int32_t *idx, double *normal_xyz, double *contact_z, double *dir_xyz,
                          double *surface_energy, size_t cap, size_t *n);

/* ---- synthetic initial conditions on the device (opt-in) ------------------------------------------
 * The recipe of the reference's generators (Cube:144-172, Pore:106-158): positions uniform per region — the cube, or
 * stacked cylinders with r = radius * sqrt(u), theta ~ U(0, 2 pi), z ~ U(z_lo, z_hi) (Pore:120-139; the caller applies
 * the argon_radius insets) — velocity components N(0, a_shape^2) (Maxwell speeds, isotropic directions; Pore:144-158),
 * path accumulators zero, flag clear.  The numbers come from Philox4x32-10 with counter (particle, block, 0, tag): they
 * depend on the seed and the particle index only, so every rank of a sharded run builds the identical system.  NOT the
 * reference's scipy / NumPy streams.  Replaces amc_upload for such runs; asynchronous on the context's stream. */
typedef struct amc_ic_config {
    int32_t struct_size;          /* sizeof(amc_ic_config)                                                       */
    int32_t n_regions;            /* 0: uniform in [0,cube_x) x [0,cube_y) x [0,cube_z); 1..8: stacked cylinders */
    uint64_t seed;
    double a_shape;               /* Maxwell scale sqrt(kT/m) (Pore:56, Cube:56)                                 */
    int64_t first[9];             /* region r holds the particles [first[r], first[r+1])                         */
    double radius[8], z_lo[8], z_hi[8];
} amc_ic_config;
int amc_init_synthetic(amc_ctx *ctx, const amc_ic_config *cfg);

/* ---- multi-GPU (one process per GPU; particles sharded by index range) ------------------------------
 * Every rank allocates all n particles but advances only its shard [lo, hi); the reference has no counterpart (its
 * parallelism is multiprocessing.Pool over cells, Pore:404-406, 546).  Per step (argon_monte_carlo_amd/dist.py):
 *   amc_mg_local            drift + walls + bounds on [lo,hi)                       (Pore:426-512 on the shard)
 *   amc_mg_pack             this rank's block -> send buffer: the positions of the shard and the list of its particles
 *                           whose velocity differs bitwise from what was last published (a collision or a wall: a
 *                           fraction of a per cent per step); the shard's particles go into the per-cell lists of the
 *                           detection grid on the way (the pack and unpack kernels see every final position once)
 *   <all-gather>            RCCL over xGMI: send of every rank -> recv (about 28 B per particle and step)
 *   amc_mg_sweep            recv -> positions (and per-cell lists) of the other shards, their velocity changes applied,
 *                           then the p-p sweep of the WHOLE system exactly as on one GPU (detect, ordered resolve):
 *                           every rank computes every collision from identical positions and velocities, so cross-shard
 *                           pairs and chains need no further exchange.  The path accumulators and the flag of a
 *                           particle matter only to its own bookkeeping: they are meaningful on the owner alone, which
 *                           is also the one that emits the particle's completed paths; the collision count is
 *                           reported by the rank that owns particle 0.  Results reach the arrays of the owner through
 *                           its next streaming pass; the other ranks learn them from the owner's next block.
 *   amc_mg_finish           bounds check after the sweep on [lo,hi), step counter, per-step counters
 * Nothing in the step waits for the host.  Energised walls (Temp:662-853) shard the same way: amc_temp_begin /
 * amc_wall_hits / amc_wall_apply act on [lo,hi); the host concatenates the hits of all ranks in rank order (=
 * ascending particle index) before drawing the random directions, so every rank consumes the two RNG streams
 * identically.  amc_mg_bounds is the bounds check between the walls and the sweep (Temp:804) on [lo,hi)
 * without a counter read-back; the sweep then runs as above. */
int amc_set_shard(amc_ctx *ctx, int64_t lo, int64_t hi);
int amc_mg_local(amc_ctx *ctx, double dt);
/* Buffers of the all-gather.  One block per rank: with m = ceil(n / world) and cap = max(4096, m / 8),
 *   float64[3][m] x|y|z of the shard (zero padded) | int64 count[16] | 16 lists of (particle, vx, vy, vz), room for
 *   cap / 16 each (the velocity changes are collected in 16 banks, one per workgroup modulo 16)
 * *block = 3 m + 16 + 4 cap float64; send holds this rank's block, recv the blocks of all ranks in rank order (the
 * output of an all-gather of `send`).  A bank that overflows in one step sets flag bit 4 (AMC_ERR_CAPACITY at the next
 * amc_mg_finish with statistics).  Shards follow the driver's rule: rank r owns base + (r < n % world)
 * particles starting at r * base + min(r, n % world), base = n / world.  Every rank uploads the same full state, which counts
 * as published (amc_upload / amc_init_synthetic on a sharded context, or amc_set_shard on an uploaded one).  The pointers stay valid until the context is
 * destroyed or the function is called with another world size. */
int amc_mg_exchange_view(amc_ctx *ctx, int world, void **send, void **recv, int64_t *block);
int amc_mg_pack(amc_ctx *ctx, int world);
int amc_mg_sweep(amc_ctx *ctx, int world, int rank);       /* world == 1: no unpack (nothing was exchanged) */
/* Detection sharded by index (the default of the driver for world > 1; amc_mg_sweep above is the replicated form): a rank
 * examines only the particles of [lo,hi), but against everybody, and keeps the pairs whose partner has the lower index —
 * every close pair of the system is found exactly once, by the owner of its higher index (the i > j rule of Pore:168-169
 * across shards).  Its pairs travel in a second, small all-gather: *block_ints 32-bit integers per rank, [0] = number of
 * pairs, [2 + 2k], [3 + 2k] = (i, j); room for max(4096, max_candidates / 4) pairs, overflow sets flag bit 0 (AMC_ERR_CAPACITY).
 *   amc_mg_pack -> all-gather #1 -> amc_mg_detect (unpack + detect) -> all-gather #2 -> amc_mg_resolve -> amc_mg_finish
 * amc_mg_resolve builds the candidate graph from the blocks of all ranks — the same on every rank — and runs the ordered
 * resolve of the whole system (replicated: it needs every member of a cluster, wherever it lives). */
int amc_mg_candidates_view(amc_ctx *ctx, int world, void **send, void **recv, int64_t *block_ints);
int amc_mg_detect(amc_ctx *ctx, int world, int rank);
int amc_mg_resolve(amc_ctx *ctx, int world);
int amc_mg_bounds(amc_ctx *ctx);
int amc_mg_finish(amc_ctx *ctx, amc_step_stats *out);      /* out == NULL: no host synchronisation (counters stay on the device) */

/* ---- measurement ----------------------------------------------------------------------------------------- */
/* With profiling on, every kernel launch is bracketed by hipEvents on the launch stream. */
#define AMC_K_DRIFT_WALLS 0
#define AMC_K_BIN_COUNT 1        /* k_bin_lists (stand-alone list build; the step driver fuses it into k_stream) and the multi-GPU pack / unpack kernels, which build the lists too */
#define AMC_K_BIN_SCAN 2         /* reserved (no such pass: the lists need neither scan nor scatter) */
#define AMC_K_BIN_SCATTER 3      /* reserved */
#define AMC_K_DETECT 4
#define AMC_K_RESOLVE 5          /* k_resolve: the ordered workgroup (entangled remainder, later rounds; no-grid mode: everything) */
#define AMC_K_BOUNDS 6
#define AMC_K_VALIDATE 7         /* reserved (validation happens inside k_clusters_wide / k_resolve)            */
#define AMC_K_RESOLVE_MORE 8     /* reserved (later rounds run inside k_resolve)                                */
#define AMC_K_COMMIT 9           /* k_commit: a sweep's commit as a launch of its own (else it rides along with the next k_stream) */
#define AMC_K_CLUSTERS_WIDE 10   /* k_clusters_wide: every small cluster emulated and validated wide, before the ordered workgroup */
#define AMC_K_FIXUP 11           /* k_fixup: joins an overlapped run's early streaming pass with the sweep's results (below) */
#define AMC_K_COUNT 12
int amc_profile(amc_ctx *ctx, int enable);
int amc_kernel_times(amc_ctx *ctx, double *total_ms /*[AMC_K_COUNT]*/, int64_t *launches /*[AMC_K_COUNT]*/);
const char *amc_kernel_name(int k);
/* amc_run overlaps the streaming pass of step s + 1 with the resolve of sweep s (whole range in one context, cube /
 * specular pore, binned detector, at least two steps; environment AMC_OVERLAP=0 turns it off, =2 runs the same kernels in
 * order on one stream).  Results are those of the plain sequence bit for bit — the loop body of Pore:416-557 / Cube:175-338
 * once per step.  out[0] = steps run that way so far, out[1] = particles a sweep pulled into a cluster after the next pass
 * had already advanced them (advanced again from the sweep's result and filed under an extra list node), out[2] = mode,
 * out[3] = extra list nodes available per step. */
int amc_overlap_stats(amc_ctx *ctx, int64_t *out /*[4]*/);

#ifdef __cplusplus
}
#endif
#endif /* ARGONMC_H */
