// amc_internal.h — host-side context of libargonmc.so and the launcher prototypes shared by its translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "amc_device.h"

#define RS_SLOT_DOUBLES 12    // x y z vx vy vz d dx dy dz flag(0/1) pad  — one slot's scratch state (96 bytes)

// ---- SoA particle state in HBM (one allocation each, float64[n]; flag uint8[n]) --------------------------------
struct amc_state {
    double *x, *y, *z, *vx, *vy, *vz, *d, *dx, *dy, *dz;
    double *px, *py, *pz;     // prior_{x,y,z}_vals — written only when keep_prior is set
    uint8_t *flag;
};

// ---- detection grid: uniform cells of edge h >= collision_range, stored per z-layer inside a square xy window ---
// (the pore's body is 68 nm wide inside a 300 nm bounding box: a dense grid would be ~12x larger than needed)
struct amc_grid {
    double x0, y0, z0, h;
    double inv_h;             // cell coordinate = floor((v - v0) * inv_h): any monotone map works as long as binning
                              // and probing use the same one
    int uniform;              // 1: every layer stores the full gx*gy window (cube) -> no table lookups
    int gx, gy, gz;           // global extent in cells
    int ncells;               // stored cells = sum over layers of lay_n^2  (0 = no grid: all-pairs mode)
    double cr_probe;          // collision_range * (1 + delta): radius of every box that is probed for list entries, and
    double cr2_probe;         // its square: the distance test against a LIST RECORD's (single-precision) position.  delta
                              // covers the rounding of the records (16 x extent x 2^-24 / collision_range, at least 1e-6)
    const int *lay_lo;        // [gz] first stored cell coordinate of the window (same for x and y)
    const int *lay_n;         // [gz] window edge in cells
    const int *lay_off;       // [gz] linear offset of the layer's first cell
};

// Per-cell particle lists of the detection grid, rebuilt every step with ONE 64-bit atomic exchange per particle:
//   head[c] = (epoch << 32) | particle   — a head whose epoch is not the current one means "empty", so nothing is ever
//                                          re-zeroed and no scan / scatter pass is needed;
//   rec[p]  = (x, y, z relative to the grid origin in SINGLE precision, next particle) — 16 bytes, written coalesced
//             (indexed by the particle itself).  The records only generate candidates and validation hits — supersets,
//             re-tested exactly on the double-precision state — so half the bytes do; the particle is filed under the
//             cell of its ROUNDED position, which the detect kernel recomputes from the same record.
struct amc_rec {
    float x, y, z;
    int next;
};
#define AMC_MARK_MULTI 0xffffffffu
struct amc_lists {
    unsigned long long *head; // [ncells]
    amc_rec *rec;             // [n + max_extra]: list NODES.  Node p < n is particle p.  Nodes n + e are handed out by the fix-up
                              // kernel of an overlapped run (amc_stream.hip) to particles that had been filed under a
                              // speculative position: their own node stays linked with a position no test passes (NaN) and
                              // the particle is filed again under node n + e, extra[e] = particle
    unsigned int epoch;       // current binning epoch (>= 1)
    int n;                    // particles (nodes below n are particles themselves)
    int *extra;               // [max_extra] node n + e -> particle (nullptr: no such node exists)
    // KEPT lists (amc_list_keep, amc_grid_dev.h): the lists of a full build live on for keep_K - 1 more steps.  A
    // particle that is still in the cell it is filed under only refreshes the position in its node; one that left gives its
    // node a position no test passes (the node stays linked: walkers pass through it) and files a NEW node n + e in its new
    // cell — one atomic exchange for the ~6 % that change cell per step in the pore instead of one per particle.  Every WAVE
    // of the streaming pass has a pool of its own (64 x (keep_K - 1) nodes: each of its particles can move once per step, so
    // it cannot run out) and a counter only it touches — no atomic, no wait: read with the state, written at the end.
    int *cell_of;             // [n] cell the particle's live node is filed under
    int *node_of;             // [n] that node
    int *wave_count;          // [waves of the streaming pass] nodes the wave has handed out since the last full build
    int wave_cap;             // nodes per wave
};
#define AMC_LIST_KEEP_DEFAULT_PORE 4      // (specular pore: 126.4 against 130.8 us per step at N = 1e6, 75.5 against 77.6 at 5e5; K = 3 the same, 6 and 8 less)
#define AMC_EXTRA_NODES(n) ((int)std::min<long long>(std::max<long long>(4096, (long long)(n) / 64), 1 << 22))

// What the streaming pass of an OVERLAPPED run needs beyond the plain one (amc_run): it reads the state the
// running sweep reads (S) and writes the other buffer (S_out), leaves out every particle the sweep's detect kernel linked
// into a candidate (graph head tagged with the sweep's epoch) and defers its events (amc_out::wev).
struct amc_ovl {
    const unsigned long long *adj_head;   // nullptr: plain pass
    unsigned int skip_epoch;              // epoch of the sweep in flight (0: none)
};

// results of the last sweep that have not been written to the particle arrays yet: the next streaming pass picks them
// up (a coalesced slot_of[] read per particle) instead of a scattered commit from one workgroup
struct amc_lazy {
    int *slot_of;
    const double *state;      // [slots][RS_SLOT_DOUBLES]
    const uint8_t *moved;
    int enabled;
};

// ---- resolve scratch ---------------------------------------------------------------------------------------------
// The resolve kernels are chains of dependent, scattered accesses by a few waves: what they cost is the number of
// memory round trips and of DISTINCT arrays (every array is a base pointer to fetch and to keep in scalar registers).
// So what one thread handles at a time is ONE record (array of structures); only what the ordered workgroup scans in
// bulk stays a plain array.
// the resolve kernels' hand-over block (mirror of rs_shared in amc_resolve_dev.h)
struct amc_resolve_ctl {
    int nslots, nedges, nhist, nev, dirty, changed, nhits, nfp, ovf, nclusters, ncomplex;
    int rounds, ncand, active, ok, edges_done;
    int lazy_ns, nslots0, hist_begin, cur_round;
};
struct rs_event {             // a completed free path found by an emulation (64 bytes), emitted at commit
    int phase, i, j, which;
    long long cell;
    int slot, pad;
    double val[4];            // total, x, y, z
};
struct amc_resolve_ws {
    // candidates: cand4[k] = (i, j, next candidate in i's list, next candidate in j's list), i > j (particle indices);
    // cand_s[k] = (slot of i, slot of j, done by the wide kernel, -)
    int4 *cand4, *cand_s;
    unsigned long long *cand_mark;  // [max_cand] (sweep epoch << 32) | successor: a later candidate of this sweep shares a particle with
                              // this one — the candidate whose exchange on the particle's graph head returned this one (detect
                              // kernel); AMC_MARK_MULTI in the low half when that happened on both of its particles
    int max_cand;
    unsigned long long *adj_head;   // [n] (sweep epoch << 32) | last candidate pushed that touches the particle
    int *slot_of;             // [n] particle -> slot or -1
    unsigned int *victim;     // [n] == sweep epoch: the sweep pulled the particle into a cluster AFTER its detect kernel (it is in
                              // no candidate, so the streaming pass of an overlapped run has advanced it speculatively)
    int max_slots;
    int4 *sl_meta;            // [max_slots] (particle, cluster label at the wide kernel's hand-over, round of the last emulation, -)
    int *sl_hits;             // collisions (low half) and failed contact solves (high half) counted on the slot: atomics only
    uint8_t *sl_moved;        // the slot's scratch state differs from the particle arrays
    double *sl_state;         // [max_slots][RS_SLOT_DOUBLES]
    int *edge_a, *edge_b;     // merge edges found by validation (particles, or slots encoded as -(slot + 2))
    int max_edges;
    double4 *hist;            // position history of the sweep: (x, y, z, slot | round << 32) per new position
    int max_hist;
    int *ov_head, *ov_next;   // overlay lists of history entries per grid cell ([ncells], [max_hist])
    // Events share the index space of the history: a hit owns the entry pair (h, h + 1) — h for particle j, h + 1 for
    // particle i — and its (at most two) events sit at the same two indices; ev_gen == 0 marks "no event"
    int *ev_gen;              // [max_hist] round of the emulation that produced the event
    rs_event *ev;             // [max_hist]
    int *ctl;                 // rs_shared in global memory: hand-over between the resolve kernels
    int *wctl;                // rs_shared of the wide cluster kernel (counters it advanced before the ordered workgroup starts)
    // (what only the ordered workgroup's large-sweep fallbacks touch comes last: the argument block is read line by line)
    int *sl_label, *sl_tmp;   // labels / sizes of the ordered workgroup when they do not fit its LDS
    uint8_t *sl_dirty;
    unsigned long long *sl_key;                    // sort keys (label<<32 | particle)
    int *order;
    double *cw_d[10];         // global fallback of the multi-particle clusters' working set (else LDS)
    int *cw_tmp, *cw_pidx, *cw_slot;
    uint8_t *cw_flag, *cw_moved;
};

// energised-wall hand-over buffers (amc_energised.hip)
struct amc_temp_ws {
    // The records live in ONE block of pinned host memory mapped into the device: the kernels write the (few hundred) hits of
    // a case straight into it and read the host's directions / energies from it — the hand-over of a case is two kernel
    // launches and two stream synchronisations, no copies.  Device addresses first, the host's view of the same bytes after.
    int *idx, *count;
    double *t, *contact, *normal, *dir, *Es, *dpz, *dE;
    unsigned char *ok;
    int *h_idx, *h_count;
    double *h_contact, *h_normal, *h_dir, *h_Es, *h_dpz, *h_dE;
    void *pin;
    int cap;
    int last_case, last_n;       // the pending amc_wall_hits
    int pre_case;                // >= 0: the hits of this case are already in the records (launched behind the previous
                                 // case's apply kernel: one synchronisation serves both)
    std::vector<int> perm;       // sorted position -> record slot of the pending hits
    // a PARKED case (amc_wall_park / amc_wall_finish: the gap case while its surface energies are still being integrated):
    // particle (-1: failed solve) and direction of every hit in record order, the energies / results of the finishing kernel
    int *def_idx;
    double *def_dir, *def_Es, *def_dpz, *def_dE;    // (the last three: device views of pinned memory, host views below)
    double *h_def_Es, *h_def_dpz, *h_def_dE;
    int def_case, def_n;
    std::vector<int> def_perm;
};

// device-RNG mode (amc_temp_cases_device): one record segment per energised case, kept until the next step
struct amc_temp_dev_ws {
    int *idx, *count;            // [7 * cap], [7]
    double *t, *contact, *normal, *dir, *Es, *dpz, *dE;
    unsigned char *ok;
    int cap;                     // records per case
    bool fetched;                // host copies below are valid for the last step
    int h_count[7];
    std::vector<int> h_idx[7];
    std::vector<double> h_dpz[7], h_dE[7];
    std::vector<unsigned char> h_ok[7];
};

struct amc_ctx {
    amc_params P;
    int device;
    hipStream_t own_stream, stream;
    std::string err;
    int64_t n, lo, hi;
    bool uploaded;
    bool keep_prior;
    amc_state S;              // the CURRENT state arrays (one of S_buf's two sets)
    amc_state S_buf[2];       // [1] is allocated by the first overlapped run: its streaming pass writes the buffer the sweep in
    char *s_slab2;            // flight does not read
    amc_grid G;
    std::vector<int> h_lay_lo, h_lay_n, h_lay_off;
    int *d_lay;               // device copy of the three layer tables, contiguous
    amc_lists B;              // the CURRENT per-cell lists (one of B_buf's two)
    amc_lists B_buf[2];
    int *extra_buf[2];        // node -> particle of the extra nodes of each list buffer, and how many were handed out
    int *extra_count;         // [2]
    int max_extra;
    amc_wev wev_buf[2];       // deferred events of the overlapped streaming pass, by step parity
    hipStream_t stream2;      // the overlapped streaming pass runs here
    hipEvent_t ev_detect, ev_stream;
    unsigned int *ovl_flags;  // two words (64 bytes apart) the streams of an overlapped run signal each other through
    unsigned int ovl_tick;    // (AMC_OVERLAP_SYNC=value: hipStreamWriteValue32 / hipStreamWaitValue32 instead of event record + wait —
    int ovl_sync_values;      // ~2 us per dependency instead of ~8, tools/ubench_xstream.hip, but the resolve suffers more: DESIGN 4.2)
    int64_t ovl_steps;        // steps run overlapped so far
    int keep_K;               // kept lists: a full build every keep_K steps (< 2: every step, the lists are not kept)
    int lists_age;            // steps since the last full build of a kept cycle (-1: the lists are not a kept cycle's)
    int lists_owner;          // who runs the cycle: 1 the streaming pass, 2 the multi-GPU exchange kernels (their node pools differ)
    int keep_threads;         // block size of the streaming pass the pools were sized for
    int overlap_mode;         // AMC_OVERLAP: 1 (default) two streams, 2 the same kernels in order on one stream (debug), 0 off
    amc_resolve_ws W;
    char *w_slab;             // the one allocation W's arrays are carved from
    char *s_slab;             // the one allocation the particle state arrays are carved from
    amc_temp_ws T;
    amc_temp_dev_ws TD;
    bool allpairs;            // no detection grid at all (single cells, N <= 4096): all-pairs detector, brute-force validation
    bool detect_ap;           // candidates come from the LDS-tiled all-pairs kernel (always without a grid; with one when detect_mode == 2)
    // outputs
    amc_out out;
    amc_path_record *d_rec;
    unsigned long long *d_hist;
    double *d_edges;
    amc_dev_counters *d_cnt;
    amc_counter_bank *d_banks;   // banked per-event counters, folded into the copy read_counters() returns
    amc_dev_counters h_prev;  // snapshot used to report per-step deltas
    long long *d_dbg;         // resolve phase timers (diagnostic, enabled by AMC_DEBUG_RESOLVE=1)
    int cw_blocks_env;        // AMC_CW_BLOCKS at creation (0 = default number of wide-kernel waves)
    // profiling
    bool profiling;
    double k_ms[AMC_K_COUNT];
    int64_t k_launches[AMC_K_COUNT];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<std::pair<int, int>> ev_pending;   // (kernel class, pool index)
    size_t ev_used;
    hipEvent_t prof_ev0, prof_ev1;  // the open bracket's events (nullptr outside a bracket / when not profiling): AMC_LAUNCH
                                    // attaches them to the dispatch itself
    // multi-GPU
    bool mg_count_pp;              // this rank adds the p-p collision count to its counters
    volatile int *h_host_ncand;    // host-mapped word written by k_resolve (candidate count of the last sweep)
    int *d_host_ncand;             // its device address
    bool lazy_pending;             // sweep results wait in the slot arrays for the next streaming pass (or amc_flush)
    bool commit_pending;           // the last sweep's commit (paths -> histograms, counters, overlay) waits for the next streaming pass (or amc_flush)
    bool commit_defer;             // ... and that sweep's results stay in the slot arrays
    unsigned int sweep_epoch;      // tag of the degree counts of the current sweep (advanced by every detect launch)
    bool plan_split;               // launch plan of the current sweep, fixed when its detect kernel is launched
    int plan_small;                // candidate pairs up to which the single resolve kernel does the whole sweep
    // pinned host staging for the small per-step read-backs (a copy into pageable memory costs ~100 us on this stack)
    char *h_pin;
    size_t h_pin_bytes;
    double *kin_send, *kin_recv;   // per-step exchange (amc_exchange.hip): one block of kin_block doubles, and world of them
    double *kin_vpub;              // [3][n] velocities as last published to the other ranks (allocated by amc_set_shard;
                                   // every upload publishes: all ranks upload the same full state)
    int kin_world;
    int64_t kin_m, kin_cap, kin_block;   // shard length (padded), capacity of the velocity-change list (all banks), 3m + banks + 4cap
    int *cand_send, *cand_recv;    // multi-GPU, detection sharded by index: this rank's candidate block ([0] count, [2 + 2k] pairs)
    int cand_cap, cand_world;      // and the blocks of all ranks (the second all-gather of a step); pairs per block
    // kept lists in the exchange kernels (pore, amc_lists): pools per wave of the pack and of the unpack kernel
    int *mg_wave_count;            // [mg_waves_pack + mg_waves_unpack]
    int mg_waves_pack, mg_waves_unpack;
    bool mg_keep;                  // the pools exist for the current world size
    int kin_mode;                  // this step's list build: 1 anew, 2 full build of a kept cycle, 3 a step in between
    size_t keep_pool;              // nodes behind the particles' own in B.rec / entries of B.extra
    bool kin_lists;                // amc_mg_pack started this step's per-cell lists (the unpack completes them)
    bool kin_counts_clear;         // the bank counters in kin_send are zero (cleared by the last unpack kernel)
};

int amc_fail(amc_ctx *c, int code, const char *fmt, ...);
#define AMC_HIP(c, call)                                                                                      \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) return amc_fail((c), AMC_ERR_HIP, "%s failed: %s (%s:%d)", #call,             \
                                               hipGetErrorString(e__), __FILE__, __LINE__);                   \
    } while (0)

// Every kernel of the step is launched through AMC_LAUNCH.  Inside an open profiling bracket the launch carries the
// bracket's two events ON THE DISPATCH (hipExtLaunchKernelGGL: start = the kernel begins to execute, stop = it has
// finished) — the same begin / end timestamps rocprofv3 --kernel-trace reports, so bench.py's per-kernel figures agree
// with the committed kernel statistics and their sum stays below the step time (events recorded around a launch with
// hipEventRecord also count the dispatch gap in front of it: 2-4 us per kernel, which made the classes of the round-2
// line add up to more than the step).  Outside a bracket both events are null: a plain launch.
#define AMC_LAUNCH(c, kernel, grid, block, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, 0, (c)->stream, (c)->prof_ev0, (c)->prof_ev1, 0, __VA_ARGS__)
#define AMC_LAUNCH_ON(c, stream_, kernel, grid, block, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, 0, stream_, (c)->prof_ev0, (c)->prof_ev1, 0, __VA_ARGS__)

// profiling brackets
void amc_prof_begin(amc_ctx *c, int kclass);
void amc_prof_end(amc_ctx *c);
void amc_prof_cancel(amc_ctx *c);
void amc_prof_collect(amc_ctx *c);

// stage bits of the streaming kernel
#define AMC_PLAN_SMALL 430      // default of amc_ctx::plan_small: measured crossover of the two launch plans (tools/plan_sweep.sh;
                                // experiments: environment variable AMC_PLAN_SMALL)
#define AMC_OVERLAP_MIN_N 300000 // default of the overlapped run (amc_api.hip): particles from which amc_run overlaps
#define AMC_ST_DRIFT 1
#define AMC_ST_WALLS 2
#define AMC_ST_BOUNDS 4
#define AMC_ST_BOUNDS_PRE 8     // the PREVIOUS step's bounds check after its sweep (Pore:550), folded into this pass

// launchers (each enqueues on c->stream; returns hipError_t of the launch)
hipError_t amc_launch_stream(amc_ctx *c, double dt, int stages, int bounds_slot, bool fuse_bin = false);
// the overlapped run (amc_stream.hip): the streaming pass of the next step while the sweep is being resolved,
// and the fix-up kernel that joins the two
hipError_t amc_launch_stream_ovl(amc_ctx *c, double dt, int stages, int from, unsigned int skip_epoch, hipStream_t stream,
                                 bool build_lists = true);
hipError_t amc_launch_fixup(amc_ctx *c, double dt, int stages, int from, unsigned int sweep_epoch);
hipError_t amc_launch_bin_ovl(amc_ctx *c, int to, unsigned int skip_epoch, hipStream_t stream);
hipError_t amc_launch_bin(amc_ctx *c);                 // stand-alone list build over all n particles (stages, multi-GPU)
hipError_t amc_launch_detect(amc_ctx *c);              // binned or all-pairs, fills W.cand_* / counters.cand_count
hipError_t amc_launch_detect_own(amc_ctx *c);          // multi-GPU: own index range against everybody, into the candidate block
hipError_t amc_launch_ingest(amc_ctx *c, int world);   // ... and the candidate graph from the gathered blocks of all ranks
hipError_t amc_launch_resolve(amc_ctx *c, bool defer_commit = false);   // resolve_A -> validate -> resolve_B -> commit
hipError_t amc_launch_apply(amc_ctx *c);                // write deferred sweep results to the particle arrays now
hipError_t amc_launch_commit(amc_ctx *c);               // the pending commit as a kernel of its own
struct amc_commit_args;
amc_commit_args amc_make_commit_args(amc_ctx *c);       // amc_stream.hip
hipError_t amc_launch_temp_hits(amc_ctx *c, int case_id);
hipError_t amc_launch_temp_apply(amc_ctx *c, int case_id, int n, bool park = false);
hipError_t amc_launch_temp_velocity(amc_ctx *c, int case_id, int n);
hipError_t amc_launch_temp_cases_device(amc_ctx *c, const amc_temp_rng *cfg);
hipError_t amc_launch_kin_pack(amc_ctx *c, int world, int rank, int unpack);
int amc_kin_banks(void);         // banks of the velocity-change list in an exchange block
