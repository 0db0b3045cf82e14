// amc_api.hip — the C ABI of libargonmc.so (include/argonmc.h): context, HBM allocation, upload/download, the
// timestep driver and measurement helpers.  No CPU compute path exists here: without a HIP device amc_create fails.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>
#include <new>
#include <type_traits>

#include "amc_host.h"

int amc_fail(amc_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

static thread_local std::string g_create_err;

// ---- profiling brackets ---------------------------------------------------------------------------------------------
void amc_prof_begin(amc_ctx *c, int kclass)
{
    if (!c->profiling) return;
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        c->ev_pool.push_back({a, b});
    }
    c->ev_pending.push_back({kclass, (int)c->ev_used});
    // the launch inside the bracket takes the two events with it (AMC_LAUNCH)
    c->prof_ev0 = c->ev_pool[c->ev_used].first;
    c->prof_ev1 = c->ev_pool[c->ev_used].second;
}
void amc_prof_end(amc_ctx *c)
{
    if (!c->profiling) return;
    c->prof_ev0 = c->prof_ev1 = nullptr;
    c->ev_used++;
    if (c->ev_used >= 4096) amc_prof_collect(c);
}
void amc_prof_cancel(amc_ctx *c)        // drop the open bracket (nothing was launched inside it)
{
    if (!c->profiling || c->ev_pending.empty()) return;
    c->ev_pending.pop_back();
    c->prof_ev0 = c->prof_ev1 = nullptr;
}
void amc_prof_collect(amc_ctx *c)
{
    if (!c->profiling || c->ev_pending.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto &pr : c->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[pr.second].first, c->ev_pool[pr.second].second) == hipSuccess) {
            c->k_ms[pr.first] += ms;
            c->k_launches[pr.first]++;
        }
    }
    c->ev_pending.clear();
    c->ev_used = 0;
}

// ---- helpers ------------------------------------------------------------------------------------------------------------
static int next_pow2(int v)
{
    int m = 1;
    while (m < v) m <<= 1;
    return m;
}

// detection grid: uniform cells, per-layer square window (full extent in the open-air end caps, gap radius inside
// the pore body); Cube: one window for all layers
static int setup_grid(amc_ctx *c)
{
    const amc_params &P = c->P;
    amc_grid &G = c->G;
    memset(&G, 0, sizeof G);
    if (c->allpairs) return AMC_OK;
    const double cr = P.collision_range;
    double xlo, xhi, zlo, zhi, volume;
    if (P.geometry == AMC_GEOM_CUBE) {
        xlo = 0; xhi = std::max(P.cube_x, P.cube_y); zlo = 0; zhi = P.cube_z;
        volume = P.cube_x * P.cube_y * P.cube_z;
    } else {
        xlo = -P.R_oa; xhi = P.R_oa; zlo = 0; zhi = P.H;
        const double pi = 3.14159265358979323846;
        volume = 2 * pi * P.R_oa * P.R_oa * P.h_oa + pi * P.R_g * P.R_g * (P.H - 2 * P.h_oa);
    }
    // list records hold positions relative to the grid origin in single precision: every test against them, and every box
    // that selects the cells to probe, is widened by 16 roundings of the largest coordinate (guard cells included)
    const double extent = 1.05 * std::max(xhi - xlo, zhi - zlo) + 8 * cr;
    const double delta = std::max(1.0e-6, 16.0 * extent * 5.9604644775390625e-08 / cr);
    const double crp = cr * (1.0 + delta);
    double h = P.fine_cell;
    if (!(h > 0)) {
        const double spacing = cbrt(volume / (double)std::max<int64_t>(1, c->n));
        // ~0.25 particles per cell: short lists (each further list element is a dependent random access) against more
        // cells per probe box; measured optimum on MI355X between 0.5 and 0.7 of the mean spacing
        h = std::max(0.63 * spacing, 2.01 * crp);
    }
    // probes look at the cells overlapped by a +-collision_range box: at most 2 per axis needs h >= 2*collision_range
    if (h < 2.00001 * crp) return amc_fail(c, AMC_ERR_INVALID, "fine_cell %g must be at least 2x the probe radius %g (collision_range %g widened for the single-precision list records)", h, crp, cr);
    // keep the table bounded
    for (;;) {
        const double nxy = ceil((xhi - xlo) / h) + 2, nz = ceil((zhi - zlo) / h) + 2;
        if (nxy * nxy * nz < 2.0e8) break;
        h *= 1.26;
    }
    G.h = h;
    G.inv_h = 1.0 / h;
    G.cr_probe = crp;
    G.cr2_probe = crp * crp;
    G.uniform = (P.geometry == AMC_GEOM_CUBE) ? 1 : 0;
    G.x0 = xlo - h; G.y0 = xlo - h; G.z0 = zlo - h;           // one guard cell on every side
    G.gx = G.gy = (int)ceil((xhi - xlo) / h) + 2;
    G.gz = (int)ceil((zhi - zlo) / h) + 2;
    c->h_lay_lo.assign(G.gz, 0);
    c->h_lay_n.assign(G.gz, G.gx);
    c->h_lay_off.assign(G.gz, 0);
    if (P.geometry != AMC_GEOM_CUBE) {
        // pore body layers only need |x|,|y| <= R_g (+ one guard cell); layers touching the end caps keep the full disc
        const double body_lo = P.h_oa + h + cr, body_hi = P.z_cold - h - cr;
        const int w_lo = std::max(0, (int)floor((-P.R_g - cr - G.x0) / h) - 1);
        const int w_hi = std::min(G.gx - 1, (int)floor((P.R_g + cr - G.x0) / h) + 1);
        for (int k = 0; k < G.gz; k++) {
            const double z_a = G.z0 + k * h, z_b = z_a + h;
            if (z_a > body_lo && z_b < body_hi) { c->h_lay_lo[k] = w_lo; c->h_lay_n[k] = w_hi - w_lo + 1; }
        }
    }
    long long off = 0;
    for (int k = 0; k < G.gz; k++) {
        c->h_lay_off[k] = (int)off;
        off += (long long)c->h_lay_n[k] * c->h_lay_n[k];
    }
    if (off > 0x7fff0000LL) return amc_fail(c, AMC_ERR_INVALID, "detection grid too large (%lld cells)", off);
    G.ncells = (int)off;
    AMC_HIP(c, dalloc(&c->d_lay, (size_t)3 * G.gz));
    AMC_HIP(c, hipMemcpy(c->d_lay, c->h_lay_lo.data(), sizeof(int) * G.gz, hipMemcpyHostToDevice));
    AMC_HIP(c, hipMemcpy(c->d_lay + G.gz, c->h_lay_n.data(), sizeof(int) * G.gz, hipMemcpyHostToDevice));
    AMC_HIP(c, hipMemcpy(c->d_lay + 2 * G.gz, c->h_lay_off.data(), sizeof(int) * G.gz, hipMemcpyHostToDevice));
    G.lay_lo = c->d_lay; G.lay_n = c->d_lay + G.gz; G.lay_off = c->d_lay + 2 * G.gz;
    return AMC_OK;
}


extern "C" {

int amc_abi_version(void) { return AMC_ABI_VERSION; }

const char *amc_last_error(const amc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

const char *amc_kernel_name(int k)
{
    static const char *names[AMC_K_COUNT] = {"drift_walls", "bin_count", "bin_scan",     "bin_scatter", "detect",  "resolve",
                                             "bounds",      "validate",  "resolve_more", "commit",      "clusters_wide", "fixup"};
    return (k >= 0 && k < AMC_K_COUNT) ? names[k] : "?";
}

void amc_destroy(amc_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->s_slab, c->s_slab2, c->d_lay, c->B_buf[0].rec, c->B_buf[0].head, c->B_buf[1].rec, c->B_buf[1].head,
                    c->extra_buf[0], c->extra_count, c->wev_buf[0].rec, c->wev_buf[0].count, c->ovl_flags,
                    c->W.ov_head, c->w_slab, c->d_rec, c->d_hist, c->d_edges, c->d_cnt, c->d_banks, c->d_dbg,
                    c->B_buf[0].cell_of, c->B_buf[0].node_of, c->B_buf[0].wave_count, c->mg_wave_count, c->keep_K >= 2 ? (void *)c->B_buf[0].extra : nullptr};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (c->h_host_ncand) hipHostFree((void *)c->h_host_ncand);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->T.pin) hipHostFree(c->T.pin);
    if (c->T.count) hipFree(c->T.count);
    if (c->T.def_idx) hipFree(c->T.def_idx);
    if (c->T.def_dir) hipFree(c->T.def_dir);
    { void *td[] = {c->TD.idx, c->TD.count, c->TD.t, c->TD.contact, c->TD.normal, c->TD.dir, c->TD.Es, c->TD.dpz, c->TD.dE, c->TD.ok};
      for (void *q : td) if (q) hipFree(q); }
    if (c->cand_send) hipFree(c->cand_send);
    if (c->cand_recv) hipFree(c->cand_recv);
    if (c->kin_send) hipFree(c->kin_send);
    if (c->kin_recv) hipFree(c->kin_recv);
    if (c->kin_vpub) hipFree(c->kin_vpub);
    for (auto &pr : c->ev_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (c->ev_detect) hipEventDestroy(c->ev_detect);
    if (c->ev_stream) hipEventDestroy(c->ev_stream);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

int amc_create(amc_ctx **out, const amc_params *p)
{
    if (!out || !p) { g_create_err = "null argument"; return AMC_ERR_INVALID; }
    *out = nullptr;
    if (p->struct_size != (int32_t)sizeof(amc_params)) {
        g_create_err = "amc_params.struct_size mismatch (ABI)";
        return AMC_ERR_INVALID;
    }
    if (p->n < 0 || p->n > 0x7fffffffLL || !(p->collision_range > 0) || !(p->argon_mass > 0) || p->geometry < 0 ||
        p->geometry > AMC_GEOM_PORE_ENERGISED) {
        g_create_err = "invalid amc_params (n, collision_range, argon_mass or geometry)";
        return AMC_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev) {
        g_create_err = "no usable HIP device (libargonmc has no CPU fallback)";
        return AMC_ERR_NO_DEVICE;
    }
    amc_ctx *c = new (std::nothrow) amc_ctx();
    if (!c) { g_create_err = "out of host memory"; return AMC_ERR_INVALID; }
    c->P = *p;
    c->device = p->device;
    c->n = p->n; c->lo = 0; c->hi = p->n;
    c->uploaded = false;
    c->keep_prior = (p->reserved0 & 1) != 0;
    c->profiling = false;
    c->ev_used = 0;
    c->prof_ev0 = c->prof_ev1 = nullptr;
    memset(c->k_ms, 0, sizeof c->k_ms);
    memset(c->k_launches, 0, sizeof c->k_launches);
    memset(&c->S, 0, sizeof c->S); memset(&c->B, 0, sizeof c->B); memset(&c->W, 0, sizeof c->W);
    memset(c->S_buf, 0, sizeof c->S_buf); memset(c->B_buf, 0, sizeof c->B_buf); memset(c->wev_buf, 0, sizeof c->wev_buf);
    c->mg_wave_count = nullptr; c->mg_waves_pack = c->mg_waves_unpack = 0; c->mg_keep = false; c->kin_mode = 1; c->keep_pool = 0;
    c->s_slab2 = nullptr; c->extra_buf[0] = c->extra_buf[1] = nullptr; c->extra_count = nullptr; c->max_extra = 0;
    c->stream2 = nullptr; c->ev_detect = c->ev_stream = nullptr;
    c->ovl_flags = nullptr; c->ovl_tick = 0;
    c->ovl_sync_values = getenv("AMC_OVERLAP_SYNC") && !strcmp(getenv("AMC_OVERLAP_SYNC"), "value");
    // AMC_OVERLAP: 1 the streaming pass of step s + 1 runs beside the resolve of sweep s inside amc_run, on a second stream;
    // 2 the same kernels in order on one stream (debugging); 0 the plain sequence (the default: measured on MI355X the resolve
    // kernels take twice as long beside the pass's memory traffic, which with the fix-up kernel and the two cross-stream
    // dependencies of a step eats what the overlap hides — DESIGN 4.2 has the numbers)
    c->overlap_mode = getenv("AMC_OVERLAP") ? atoi(getenv("AMC_OVERLAP")) : 0;
    c->ovl_steps = 0;
    c->T.idx = nullptr; c->T.count = nullptr; c->T.t = c->T.contact = c->T.normal = c->T.dir = c->T.Es = c->T.dpz = c->T.dE = nullptr;
    c->T.ok = nullptr; c->T.cap = 0; c->T.last_case = -1; c->T.last_n = 0; c->T.pre_case = -1; c->T.pin = nullptr;
    c->T.def_idx = nullptr; c->T.def_dir = nullptr; c->T.def_case = -1; c->T.def_n = 0;
    memset(&c->out, 0, sizeof c->out); memset(&c->h_prev, 0, sizeof c->h_prev);
    c->d_lay = nullptr; c->d_banks = nullptr; c->d_rec = nullptr; c->d_hist = nullptr; c->d_edges = nullptr;
    c->d_dbg = nullptr; c->w_slab = nullptr; c->s_slab = nullptr;
    c->cw_blocks_env = getenv("AMC_CW_BLOCKS") ? atoi(getenv("AMC_CW_BLOCKS")) : 0;
    c->mg_count_pp = true;
    c->lazy_pending = false; c->commit_pending = false; c->commit_defer = false;
    c->h_host_ncand = nullptr; c->d_host_ncand = nullptr;
    c->d_cnt = nullptr; c->own_stream = nullptr; c->h_pin = nullptr; c->h_pin_bytes = 0; c->plan_split = false;
    c->plan_small = AMC_PLAN_SMALL;
    if (const char *e = getenv("AMC_PLAN_SMALL")) { const int v = atoi(e); if (v >= 0) c->plan_small = v; }
    c->cand_send = c->cand_recv = nullptr; c->cand_cap = 0; c->cand_world = 0;
    c->kin_send = c->kin_recv = c->kin_vpub = nullptr; c->kin_world = 0; c->kin_m = c->kin_cap = c->kin_block = 0; c->kin_lists = false; c->kin_counts_clear = false;
    c->TD.idx = nullptr; c->TD.count = nullptr; c->TD.t = c->TD.contact = c->TD.normal = c->TD.dir = c->TD.Es = c->TD.dpz = c->TD.dE = nullptr;
    c->TD.ok = nullptr; c->TD.cap = 0; c->TD.fetched = false;
    c->stream = nullptr;
    int rc = AMC_OK;
#define CK(call)                                                                                       \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            rc = amc_fail(c, AMC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__));             \
            goto fail;                                                                                 \
        }                                                                                              \
    } while (0)
    {
        CK(hipSetDevice(c->device));
        CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        const size_t n = (size_t)c->n;
        // detection mode: only single cells (AMC_GEOM_CELL: no geometry to lay a grid over) run without the detection grid —
        // all-pairs detector, brute-force validation, everything in the one ordered workgroup.  Cube and pore contexts use
        // the grid and the wide cluster kernel at ANY size: measured at N = 4,096 the grid path takes 32 us per step, the
        // gridless one 132 (N = 1,000: 28 against 45; round 2 sent N <= 4,096 down the gridless path).  AMC_ALLPAIRS_MAX_N
        // restores a threshold for experiments.
        // (detect_mode 2: the all-pairs DETECTOR in front of the same grid-based resolve — the kernel the reference's
        // pairwise loop maps to directly, measurable at full size against the fp64 vector peak)
        long long small_n = 0;
        if (const char *e = getenv("AMC_ALLPAIRS_MAX_N")) small_n = atoll(e);
        c->allpairs = (p->geometry == AMC_GEOM_CELL) || (p->detect_mode != 1 && c->n <= small_n);
        c->detect_ap = c->allpairs || p->detect_mode == 2;
        if (p->geometry == AMC_GEOM_CELL && p->detect_mode == 1) {
            rc = amc_fail(c, AMC_ERR_INVALID, "AMC_GEOM_CELL has no cell grid: detect_mode must be 0 or 2");
            goto fail;
        }
        double **st[] = {&c->S.x, &c->S.y, &c->S.z, &c->S.vx, &c->S.vy, &c->S.vz, &c->S.d, &c->S.dx, &c->S.dy, &c->S.dz,
                         &c->S.px, &c->S.py, &c->S.pz};
        {
            // the particle state in ONE allocation: the resolve kernels gather the eleven fields of a particle from eleven
            // arrays — in one slab they share address-translation entries instead of needing one each
            const size_t per = ((sizeof(double) * std::max<size_t>(n, 1)) + 255) & ~(size_t)255;
            const size_t total = 13 * per + ((std::max<size_t>(n, 1) + 255) & ~(size_t)255);
            CK(hipMalloc((void **)&c->s_slab, total));
            CK(hipMemsetAsync(c->s_slab, 0, total, c->stream));
            size_t off = 0;
            for (auto pp : st) { *pp = (double *)(c->s_slab + off); off += per; }
            c->S.flag = (uint8_t *)(c->s_slab + off);
            c->S_buf[0] = c->S;
        }
        if ((rc = setup_grid(c)) != AMC_OK) goto fail;
        if (!c->allpairs) {
            const size_t nc = (size_t)c->G.ncells;
            c->max_extra = AMC_EXTRA_NODES(c->n);
            // kept lists (amc_lists): AMC_LIST_KEEP=K, a full build every K steps.  Off in an overlapped run (its fix-up kernel
            // files particles itself) and when the all-pairs detector is in front.  (The energised pore files its particles in
            // the bounds pass that follows the wall cases, amc_temp_end: the same pass, the same cycle.)
            c->keep_K = 0; c->lists_age = -1; c->lists_owner = 0; c->keep_threads = 0;
            size_t pool = 0, keep_waves = 0;
            {
                int K = (p->geometry == AMC_GEOM_PORE || p->geometry == AMC_GEOM_PORE_ENERGISED) ? AMC_LIST_KEEP_DEFAULT_PORE : 0;
                if (const char *e = getenv("AMC_LIST_KEEP")) K = atoi(e);
                if (c->overlap_mode || c->detect_ap || p->geometry == AMC_GEOM_CELL) K = 0;
                const int threads = getenv("AMC_STREAM_BS") ? atoi(getenv("AMC_STREAM_BS")) : 256;
                const long long nwaves = (((long long)n + threads - 1) / threads) * (threads / 64);
                // (a wave's pool holds everything its 64 particles could hand out in K - 1 steps; shorter cycles rather than more
                // than 2^30 nodes)
                while (K >= 2 && (long long)n + nwaves * 64 * (K - 1) > 0x3fffffffLL) K--;
                if (K >= 2 && n > 0 && threads % 64 == 0) {
                    c->keep_K = K; c->keep_threads = threads;
                    c->B.wave_cap = 64 * (K - 1);
                    pool = (size_t)c->B.wave_cap * (size_t)nwaves;
                    keep_waves = (size_t)nwaves;
                }
            }
            c->keep_pool = pool;
            CK(dalloc(&c->B.rec, n + std::max((size_t)c->max_extra, pool)));
            CK(dalloc(&c->B.head, nc + 1));
            CK(hipMemsetAsync(c->B.head, 0, sizeof(unsigned long long) * (nc + 1), c->stream));
            c->B.epoch = 0;
            c->B.n = (int)c->n; c->B.extra = nullptr;
            c->B.cell_of = c->B.node_of = c->B.wave_count = nullptr;
            if (c->keep_K >= 2) {
                CK(dalloc(&c->B.extra, pool));
                CK(dalloc(&c->B.cell_of, n));
                CK(dalloc(&c->B.node_of, n));
                CK(dalloc(&c->B.wave_count, keep_waves));
                CK(hipMemsetAsync(c->B.wave_count, 0, sizeof(int) * keep_waves, c->stream));
            } else {
                c->B.wave_cap = 0;
            }
            c->B_buf[0] = c->B;
            CK(dalloc(&c->W.ov_head, nc));
            CK(hipMemsetAsync(c->W.ov_head, 0xff, sizeof(int) * std::max<size_t>(nc, 1), c->stream));
        }
        // resolve work space
        amc_resolve_ws &W = c->W;
        long long mc = p->max_candidates > 0 ? p->max_candidates : std::max<long long>(4096, c->n / 8 + 1024);
        if (mc > 0x3fffffff) mc = 0x3fffffff;
        W.max_cand = (int)mc;
        // every candidate brings two slots of its own (2k, 2k + 1); particles that join a cluster later take theirs from a counter
        W.max_slots = (int)std::min<long long>(2 * mc + std::max<long long>(1024, mc / 2), 0x7ffffff0LL);
        W.max_edges = 4 * W.max_slots + 1024;
        W.max_hist = 8 * W.max_slots + 1024;
        // ONE allocation for the whole sweep work space: the resolve kernels are chains of dependent, scattered accesses to
        // some sixty small arrays — carved from one slab they share a handful of translation entries instead of one each
        {
            const size_t ms = (size_t)W.max_slots;
            auto carve = [&](char *base) -> size_t {
                size_t off = 0;
                auto take = [&](auto **pp, size_t count) {
                    using T = std::remove_pointer_t<std::remove_pointer_t<decltype(pp)>>;
                    off = (off + 255) & ~(size_t)255;
                    *pp = base ? (T *)(base + off) : nullptr;
                    off += sizeof(T) * std::max<size_t>(count, 1);
                };
                take(&W.ctl, 64); take(&W.wctl, 64);
                take(&W.cand4, (size_t)W.max_cand); take(&W.cand_s, (size_t)W.max_cand); take(&W.cand_mark, (size_t)W.max_cand);
                take(&W.sl_meta, ms); take(&W.sl_hits, ms); take(&W.sl_moved, ms);
                take(&W.sl_state, (size_t)RS_SLOT_DOUBLES * ms);
                take(&W.sl_label, ms); take(&W.sl_tmp, ms); take(&W.sl_dirty, ms); take(&W.order, ms);
                take(&W.sl_key, (size_t)next_pow2(W.max_slots));
                for (int k = 0; k < 10; k++) take(&W.cw_d[k], ms);
                take(&W.cw_tmp, ms); take(&W.cw_pidx, ms); take(&W.cw_slot, ms); take(&W.cw_flag, ms); take(&W.cw_moved, ms);
                take(&W.edge_a, (size_t)W.max_edges); take(&W.edge_b, (size_t)W.max_edges);
                take(&W.hist, (size_t)W.max_hist); take(&W.ov_next, (size_t)W.max_hist);
                take(&W.ev_gen, (size_t)W.max_hist); take(&W.ev, (size_t)W.max_hist);
                take(&W.adj_head, n); take(&W.slot_of, n); take(&W.victim, n);
                return (off + 255) & ~(size_t)255;
            };
            const size_t total = carve(nullptr);
            CK(hipMalloc((void **)&c->w_slab, total));
            CK(hipMemsetAsync(c->w_slab, 0, total, c->stream));
            carve(c->w_slab);
        }
        // AMC_MAX_HIST (diagnostic): the history / overlay entries a sweep may USE (the allocation keeps its size): lets a test
        // drive the wide kernel's overlay protocol into its capacity limit at sizes the oracle handles in seconds
        if (const char *e = getenv("AMC_MAX_HIST")) { const int v = atoi(e); if (v > 0 && v < W.max_hist) W.max_hist = v; }
        { amc_resolve_ctl z; memset(&z, 0, sizeof z); z.cur_round = 1; CK(hipMemcpyAsync(W.wctl, &z, sizeof z, hipMemcpyHostToDevice, c->stream)); CK(hipStreamSynchronize(c->stream)); }
        c->sweep_epoch = 0;
        CK(hipMemsetAsync(W.slot_of, 0xff, sizeof(int) * std::max<size_t>(n, 1), c->stream));
        // outputs
        long long mp = p->max_paths > 0 ? p->max_paths : (p->max_paths < 0 ? 0 : (1LL << 20));   // < 0: histograms only
        if (mp > 0x7fffffff) mp = 0x7fffffff;
        if (mp > 0) CK(dalloc(&c->d_rec, (size_t)mp));
        CK(dalloc(&c->d_cnt, 1));
        CK(hipMemsetAsync(c->d_cnt, 0, sizeof(amc_dev_counters), c->stream));
        CK(dalloc(&c->d_banks, AMC_COUNTER_BANKS));
        CK(hipMemsetAsync(c->d_banks, 0, sizeof(amc_counter_bank) * AMC_COUNTER_BANKS, c->stream));
        c->out.banks = c->d_banks;
        c->out.rec = c->d_rec; c->out.cap = (unsigned)mp; c->out.cnt = c->d_cnt; c->out.step = 0;
        c->out.nbins = 0; c->out.hist = nullptr; c->out.edges = nullptr; c->out.lo = p->hist_lo; c->out.hi = p->hist_hi;
        if (p->hist_bins > 0 && p->hist_hi > p->hist_lo) {
            const int nb = p->hist_bins;
            CK(dalloc(&c->d_hist, (size_t)4 * nb * AMC_COUNTER_BANKS));
            CK(hipMemsetAsync(c->d_hist, 0, sizeof(unsigned long long) * 4 * nb * AMC_COUNTER_BANKS, c->stream));
            CK(dalloc(&c->d_edges, (size_t)nb + 1));
            // np.linspace(lo, hi, nb+1): start + k*step with step = (hi-lo)/nb, last element forced to hi
            std::vector<double> ed(nb + 1);
            const double step = (p->hist_hi - p->hist_lo) / (double)nb;
            for (int k = 0; k <= nb; k++) ed[k] = p->hist_lo + (double)k * step;
            ed[nb] = p->hist_hi;
            CK(hipMemcpy(c->d_edges, ed.data(), sizeof(double) * (nb + 1), hipMemcpyHostToDevice));
            c->out.nbins = nb; c->out.hist = c->d_hist; c->out.edges = c->d_edges; c->out.bin_step = step;
        }
        {
            void *hp = nullptr, *dp = nullptr;
            if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                c->h_host_ncand = (volatile int *)hp;
                *c->h_host_ncand = 0;
                c->d_host_ncand = (int *)dp;
            }
        }
        c->h_pin_bytes = (size_t)4 << 20;
        if (hipHostMalloc((void **)&c->h_pin, c->h_pin_bytes, hipHostMallocDefault) != hipSuccess) { c->h_pin = nullptr; c->h_pin_bytes = 0; }
        if (getenv("AMC_DEBUG_RESOLVE")) {
            CK(dalloc(&c->d_dbg, 128 + 128 * 512));
            CK(hipMemsetAsync(c->d_dbg, 0, sizeof(long long) * (128 + 128 * 512), c->stream));
            { const long long big = 0x7fffffffffffffffLL; CK(hipMemcpyAsync(c->d_dbg + 28, &big, sizeof big, hipMemcpyHostToDevice, c->stream)); }
        }
        CK(hipStreamSynchronize(c->stream));
    }
#undef CK
    *out = c;
    return AMC_OK;
fail:
    g_create_err = c->err;
    amc_destroy(c);
    return rc;
}

int amc_set_stream(amc_ctx *c, void *hip_stream)
{
    if (!c) return AMC_ERR_INVALID;
    hipSetDevice(c->device);
    amc_prof_collect(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return AMC_OK;
}

int amc_use_null_stream(amc_ctx *c)
{
    if (!c) return AMC_ERR_INVALID;
    hipSetDevice(c->device);
    amc_prof_collect(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    c->stream = nullptr;        // HIP's NULL stream
    return AMC_OK;
}

int amc_synchronize(amc_ctx *c)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    return AMC_OK;
}

int amc_upload(amc_ctx *c, const double *x, const double *y, const double *z, const double *vx, const double *vy,
               const double *vz, const double *dist, const double *dist_x, const double *dist_y, const double *dist_z,
               const uint8_t *full_path)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    c->lists_age = -1;          // (kept lists: a new state starts with a full build)
    const size_t nb = sizeof(double) * (size_t)c->n;
    const double *src[] = {x, y, z, vx, vy, vz, dist, dist_x, dist_y, dist_z};
    double *dst[] = {c->S.x, c->S.y, c->S.z, c->S.vx, c->S.vy, c->S.vz, c->S.d, c->S.dx, c->S.dy, c->S.dz};
    for (int k = 0; k < 10; k++)
        if (src[k] && nb) AMC_HIP(c, hipMemcpyAsync(dst[k], src[k], nb, hipMemcpyHostToDevice, c->stream));
    if (full_path && c->n) AMC_HIP(c, hipMemcpyAsync(c->S.flag, full_path, (size_t)c->n, hipMemcpyHostToDevice, c->stream));
    { int rc_ = amc_publish_velocities(c); if (rc_) return rc_; }
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    c->uploaded = true;
    return AMC_OK;
}

int amc_publish_velocities(amc_ctx *c)
{
    if (!c->kin_vpub || c->n <= 0) return AMC_OK;       // not a sharded context
    const size_t nb = sizeof(double) * (size_t)c->n;
    const double *src[3] = {c->S.vx, c->S.vy, c->S.vz};
    for (int e = 0; e < 3; e++)
        AMC_HIP(c, hipMemcpyAsync(c->kin_vpub + (size_t)e * (size_t)c->n, src[e], nb, hipMemcpyDeviceToDevice, c->stream));
    return AMC_OK;
}

int amc_download(amc_ctx *c, double *x, double *y, double *z, double *vx, double *vy, double *vz, double *dist,
                 double *dist_x, double *dist_y, double *dist_z, uint8_t *full_path)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    const size_t nb = sizeof(double) * (size_t)c->n;
    double *dst[] = {x, y, z, vx, vy, vz, dist, dist_x, dist_y, dist_z};
    double *src[] = {c->S.x, c->S.y, c->S.z, c->S.vx, c->S.vy, c->S.vz, c->S.d, c->S.dx, c->S.dy, c->S.dz};
    for (int k = 0; k < 10; k++)
        if (dst[k] && nb) AMC_HIP(c, hipMemcpyAsync(dst[k], src[k], nb, hipMemcpyDeviceToHost, c->stream));
    if (full_path && c->n) AMC_HIP(c, hipMemcpyAsync(full_path, c->S.flag, (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    return AMC_OK;
}

int amc_download_prior(amc_ctx *c, double *px, double *py, double *pz)
{
    if (!c) return AMC_ERR_INVALID;
    if (!c->keep_prior) return amc_fail(c, AMC_ERR_STATE, "prior_*_vals are only kept when amc_params.reserved0 bit0 is set");
    AMC_HIP(c, hipSetDevice(c->device));
    const size_t nb = sizeof(double) * (size_t)c->n;
    if (px && nb) AMC_HIP(c, hipMemcpyAsync(px, c->S.px, nb, hipMemcpyDeviceToHost, c->stream));
    if (py && nb) AMC_HIP(c, hipMemcpyAsync(py, c->S.py, nb, hipMemcpyDeviceToHost, c->stream));
    if (pz && nb) AMC_HIP(c, hipMemcpyAsync(pz, c->S.pz, nb, hipMemcpyDeviceToHost, c->stream));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    return AMC_OK;
}

// ---- the step ----------------------------------------------------------------------------------------------------------
static void fold_banks(amc_dev_counters *h, const amc_counter_bank *b)
{
    for (int k = 0; k < AMC_COUNTER_BANKS; k++) {
        h->n_wall += b[k].n_wall; h->n_paths += b[k].n_paths; h->n_paths_total += b[k].n_paths_total;
        h->n_fp_errors += b[k].n_fp_errors; h->n_pp += b[k].n_pp;
    }
}

int amc_read_counters(amc_ctx *c, amc_dev_counters *h)
{
    if (c->commit_pending) {            // the last sweep's paths / counters are not in yet: commit it now (its results stay deferred)
        AMC_HIP(c, amc_launch_commit(c));
        c->commit_pending = false;
    }
    amc_counter_bank banks[AMC_COUNTER_BANKS];
    amc_stage st(c);
    AMC_HIP(c, st.get(h, c->d_cnt, sizeof *h));
    AMC_HIP(c, st.get(banks, c->d_banks, sizeof banks));
    AMC_HIP(c, st.finish());
    fold_banks(h, banks);
    return AMC_OK;
}

static void delta_stats(const amc_dev_counters &now, const amc_dev_counters &prev, amc_step_stats *o)
{
    o->n_pp = (int64_t)(now.n_pp - prev.n_pp);
    o->n_wall = (int64_t)(now.n_wall - prev.n_wall);
    o->n_oob_walls = (int64_t)(now.n_oob_walls - prev.n_oob_walls);
    o->n_oob_pp = (int64_t)(now.n_oob_pp - prev.n_oob_pp);
    o->n_paths = (int64_t)(now.n_paths - prev.n_paths);
    o->n_candidates = (int64_t)(now.n_candidates - prev.n_candidates);
    o->n_clusters = (int64_t)(now.n_clusters - prev.n_clusters);
    o->n_rounds = (int64_t)(now.n_rounds - prev.n_rounds);
    o->n_fp_errors = (int64_t)(now.n_fp_errors - prev.n_fp_errors);
    o->flags = (int64_t)now.flags;
}

int amc_finish_stats(amc_ctx *c, amc_step_stats *out)
{
    amc_dev_counters now;
    int rc = amc_read_counters(c, &now);
    if (rc) return rc;
    amc_step_stats st;
    delta_stats(now, c->h_prev, &st);
    c->h_prev = now;
    if (out) *out = st;
    if (now.flags & 21ULL) {    // candidate / resolve work-space / velocity-change list overflow; a full path-record buffer (bit1) only stops recording
        const unsigned long long f = now.flags;
        // clear the sticky flags on the device so that a later call can succeed after the caller drained / resized
        unsigned long long zero = 0;
        hipMemcpyAsync(&c->d_cnt->flags, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream);
        hipStreamSynchronize(c->stream);
        c->h_prev.flags = 0;
        return amc_fail(c, AMC_ERR_CAPACITY, "device work buffer overflow (flags=%llu: 1 candidates, 4 resolve work space, 16 velocity changes of one step in the multi-GPU exchange)", f);
    }
    if (st.n_fp_errors > 0 && c->P.geometry != AMC_GEOM_PORE_ENERGISED && !(c->P.reserved1 & 1))
        return amc_fail(c, AMC_ERR_FP, "%lld event(s) where the reference raises FloatingPointError", (long long)st.n_fp_errors);
    return AMC_OK;
}

// sweep results deferred to the next streaming pass: write them now (before anything else reads the particle arrays)
int amc_flush(amc_ctx *c)
{
    if (c->commit_pending) {            // (before the results are applied: the commit leaves the number of deferred slots)
        AMC_HIP(c, amc_launch_commit(c));
        c->commit_pending = false;
    }
    if (!c->lazy_pending) return AMC_OK;
    AMC_HIP(c, amc_launch_apply(c));
    c->lazy_pending = false;
    return AMC_OK;
}

int amc_enqueue_sweep(amc_ctx *c, bool counted, bool defer_commit)
{
    if (!counted) AMC_HIP(c, amc_launch_bin(c));
    AMC_HIP(c, amc_launch_detect(c));
    AMC_HIP(c, amc_launch_resolve(c, defer_commit));
    if (defer_commit) c->lazy_pending = true;
    return AMC_OK;
}


// fold_prev_bounds: the previous step of the same amc_run left its post-sweep bounds check to this step's streaming
// pass; defer_bounds: leave this step's to the next one (the caller runs it separately after the last step)
static int enqueue_step(amc_ctx *c, double dt, bool fold_prev_bounds = false, bool defer_bounds = false)
{
    const int g = c->P.geometry;
    int rc;
    if (g == AMC_GEOM_CELL) {
        if ((rc = amc_enqueue_sweep(c))) return rc;
    } else if (g == AMC_GEOM_CUBE || g == AMC_GEOM_PORE) {
        // the streaming pass also counts the particles into the detection grid when it covers all of them
        const bool fuse = !c->allpairs && c->lo == 0 && c->hi == c->n;
        int st = (g == AMC_GEOM_CUBE) ? (AMC_ST_DRIFT | AMC_ST_WALLS) : (AMC_ST_DRIFT | AMC_ST_WALLS | AMC_ST_BOUNDS);
        if (fold_prev_bounds && g == AMC_GEOM_PORE) st |= AMC_ST_BOUNDS_PRE;
        AMC_HIP(c, amc_launch_stream(c, dt, st, 0, fuse));
        // the scattered commit is deferred: the next streaming pass over all particles (the bounds check for the pore,
        // the next step's drift for the cube) picks the results up through slot_of[]
        if ((rc = amc_enqueue_sweep(c, fuse, c->lo == 0 && c->hi == c->n))) return rc;
        if (g == AMC_GEOM_PORE && !defer_bounds) AMC_HIP(c, amc_launch_stream(c, dt, AMC_ST_BOUNDS, 1));
    } else {
        return amc_fail(c, AMC_ERR_INVALID, "energised walls need the host handshake: use the Python driver (amc_wall_hits/apply)");
    }
    c->out.step++;
    return AMC_OK;
}

int amc_timestep(amc_ctx *c, double dt, amc_step_stats *out)
{
    if (!c) return AMC_ERR_INVALID;
    if (!c->uploaded) return amc_fail(c, AMC_ERR_STATE, "amc_timestep before amc_upload");
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_step(c, dt);
    if (rc) return rc;
    return amc_finish_stats(c, out);
}

// ---- the overlapped run -----------------------------------------------------------------------------------
// Resolving a sweep is latency-bound work for a few hundred waves (k_clusters_wide, the ordered workgroup) and needs the
// PRE-sweep state; the next step's streaming pass is bandwidth- and atomic-bound work for the whole chip and needs the sweep's
// results only for the few thousand particles it touches.  Inside amc_run the two therefore run side by side: the pass reads
// the state buffer the resolve reads and writes the other one, leaves out the particles of the sweep's candidates, and a
// fix-up kernel advances those from the sweep's results afterwards (amc_stream.hip).  What is needed for it — a second set of
// state arrays and per-cell lists, the deferred-event buffers, a second stream — is allocated by the first such run.
static int ensure_overlap(amc_ctx *c)
{
    if (c->s_slab2) return AMC_OK;
    const size_t n = (size_t)std::max<int64_t>(c->n, 1);
    {
        const size_t per = ((sizeof(double) * n) + 255) & ~(size_t)255;
        const size_t total = 10 * per + ((n + 255) & ~(size_t)255);
        AMC_HIP(c, hipMalloc((void **)&c->s_slab2, total));
        AMC_HIP(c, hipMemsetAsync(c->s_slab2, 0, total, c->stream));
        amc_state &T = c->S_buf[1];
        double **st[] = {&T.x, &T.y, &T.z, &T.vx, &T.vy, &T.vz, &T.d, &T.dx, &T.dy, &T.dz};
        size_t off = 0;
        for (auto pp : st) { *pp = (double *)(c->s_slab2 + off); off += per; }
        T.flag = (uint8_t *)(c->s_slab2 + off);
        T.px = c->S_buf[0].px; T.py = c->S_buf[0].py; T.pz = c->S_buf[0].pz;      // (prior_*_vals are not kept by these runs)
    }
    {
        const size_t nc = (size_t)c->G.ncells;
        amc_lists &T = c->B_buf[1];
        AMC_HIP(c, dalloc(&T.rec, n + (size_t)c->max_extra));
        AMC_HIP(c, dalloc(&T.head, nc + 1));
        AMC_HIP(c, hipMemsetAsync(T.head, 0, sizeof(unsigned long long) * (nc + 1), c->stream));
        T.epoch = 0; T.n = (int)c->n;
        AMC_HIP(c, dalloc(&c->extra_buf[0], (size_t)2 * c->max_extra));
        c->extra_buf[1] = c->extra_buf[0] + c->max_extra;
        AMC_HIP(c, dalloc(&c->extra_count, 2));
        AMC_HIP(c, hipMemsetAsync(c->extra_count, 0, 2 * sizeof(int), c->stream));
    }
    {
        // deferred events of a pass: wall hits (~1e-3 per particle and step in the pore), by bank
        const int cap = (int)std::max<long long>(1024, (long long)c->n / (4 * AMC_COUNTER_BANKS));
        AMC_HIP(c, dalloc(&c->wev_buf[0].rec, (size_t)2 * AMC_COUNTER_BANKS * cap));
        AMC_HIP(c, dalloc(&c->wev_buf[0].count, (size_t)2 * AMC_COUNTER_BANKS));
        AMC_HIP(c, hipMemsetAsync(c->wev_buf[0].count, 0, sizeof(unsigned int) * 2 * AMC_COUNTER_BANKS, c->stream));
        c->wev_buf[0].cap = c->wev_buf[1].cap = cap;
        c->wev_buf[1].rec = c->wev_buf[0].rec + (size_t)AMC_COUNTER_BANKS * cap;
        c->wev_buf[1].count = c->wev_buf[0].count + AMC_COUNTER_BANKS;
    }
    AMC_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    {
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) != hipSuccess || !can) c->ovl_sync_values = 0;
        AMC_HIP(c, dalloc(&c->ovl_flags, 64));
        AMC_HIP(c, hipMemsetAsync(c->ovl_flags, 0, 64 * sizeof(unsigned int), c->stream));
    }
    AMC_HIP(c, hipEventCreateWithFlags(&c->ev_detect, hipEventDisableTiming));
    AMC_HIP(c, hipEventCreateWithFlags(&c->ev_stream, hipEventDisableTiming));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    return AMC_OK;
}

static int run_overlapped(amc_ctx *c, double dt, int64_t nsteps)
{
    int rc = ensure_overlap(c);
    if (rc) return rc;
    if ((rc = amc_flush(c))) return rc;                 // the state is complete in the current arrays
    const int g = c->P.geometry;
    const bool two = c->overlap_mode == 1;
    int cur = (c->S.x == c->S_buf[1].x) ? 1 : 0;
    c->B_buf[cur] = c->B;                               // (the list epoch lives in the current copy)
    c->B_buf[0].extra = c->extra_buf[0]; c->B_buf[1].extra = c->extra_buf[1];
    unsigned int prev_epoch = 0;                        // the sweep in flight (none before the first step)
    for (int64_t s = 0; s < nsteps; s++) {
        int st = (g == AMC_GEOM_CUBE) ? (AMC_ST_DRIFT | AMC_ST_WALLS) : (AMC_ST_DRIFT | AMC_ST_WALLS | AMC_ST_BOUNDS);
        if (g == AMC_GEOM_PORE && s > 0) st |= AMC_ST_BOUNDS_PRE;       // Pore:550 of the previous step
        hipStream_t sp = two ? c->stream2 : c->stream;
        if (two) {
            // the pass may start once the previous sweep's detect kernel has marked its candidates' particles (and, at the
            // first step, once everything queued before the run has finished)
            c->ovl_tick++;
            if (c->ovl_sync_values) {
                AMC_HIP(c, hipStreamWriteValue32(c->stream, c->ovl_flags, c->ovl_tick, 0));
                AMC_HIP(c, hipStreamWaitValue32(c->stream2, c->ovl_flags, c->ovl_tick, hipStreamWaitValueGte, 0xffffffffu));
            } else {
                AMC_HIP(c, hipEventRecord(c->ev_detect, c->stream));
                AMC_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_detect, 0));
            }
        }
        if (s > 0) {
            // (recorded above, BEHIND the detect kernel of step s - 1 and in front of its resolve kernels, which follow here)
            AMC_HIP(c, amc_launch_resolve(c, true));
        }
        static const bool split = getenv("AMC_OVERLAP_SPLIT") && atoi(getenv("AMC_OVERLAP_SPLIT")) != 0;    // (experiment)
        AMC_HIP(c, amc_launch_stream_ovl(c, dt, st, cur, prev_epoch, sp, !split));
        if (split) AMC_HIP(c, amc_launch_bin_ovl(c, 1 - cur, prev_epoch, sp));
        if (two) {
            if (c->ovl_sync_values) {
                AMC_HIP(c, hipStreamWriteValue32(c->stream2, c->ovl_flags + 16, c->ovl_tick, 0));
                AMC_HIP(c, hipStreamWaitValue32(c->stream, c->ovl_flags + 16, c->ovl_tick, hipStreamWaitValueGte, 0xffffffffu));
            } else {
                AMC_HIP(c, hipEventRecord(c->ev_stream, c->stream2));
                AMC_HIP(c, hipStreamWaitEvent(c->stream, c->ev_stream, 0));
            }
        }
        AMC_HIP(c, amc_launch_fixup(c, dt, st, cur, prev_epoch));
        c->commit_pending = false; c->lazy_pending = false;             // (consumed by the fix-up kernel)
        cur = 1 - cur;
        c->S = c->S_buf[cur];
        c->B = c->B_buf[cur];
        AMC_HIP(c, amc_launch_detect(c));
        prev_epoch = c->sweep_epoch;
        c->out.step++;
        c->ovl_steps++;
    }
    // the last sweep: resolved, and left to the plain machinery (its commit and its results wait for the next streaming pass,
    // a flush or a read of the counters, as after any step)
    AMC_HIP(c, amc_launch_resolve(c, true));
    c->lazy_pending = true;
    c->B_buf[cur] = c->B;
    // the current lists' extra nodes die with them (the next build starts from the particles' own nodes)
    AMC_HIP(c, hipMemsetAsync(c->extra_count, 0, 2 * sizeof(int), c->stream));
    AMC_HIP(c, hipMemsetAsync(c->wev_buf[0].count, 0, sizeof(unsigned int) * 2 * AMC_COUNTER_BANKS, c->stream));
    if (g == AMC_GEOM_PORE) AMC_HIP(c, amc_launch_stream(c, dt, AMC_ST_BOUNDS, 1));     // Pore:550 of the last step
    return AMC_OK;
}

int amc_run(amc_ctx *c, double dt, int64_t nsteps, amc_step_stats *sum)
{
    if (!c) return AMC_ERR_INVALID;
    if (!c->uploaded) return amc_fail(c, AMC_ERR_STATE, "amc_run before amc_upload");
    AMC_HIP(c, hipSetDevice(c->device));
    const bool whole = c->lo == 0 && c->hi == c->n && !c->allpairs;
    if (c->overlap_mode && whole && nsteps >= 2 && !c->keep_prior && !c->detect_ap && c->n > 0 &&
        (c->P.geometry == AMC_GEOM_CUBE || c->P.geometry == AMC_GEOM_PORE)) {
        int rc = run_overlapped(c, dt, nsteps);
        if (rc) return rc;
        return amc_finish_stats(c, sum);
    }
    // inside the run only the last step needs its own post-sweep bounds pass (needs the whole range in one context)
    const bool fold = c->P.geometry == AMC_GEOM_PORE && whole;
    for (int64_t s = 0; s < nsteps; s++) {
        int rc = enqueue_step(c, dt, fold && s > 0, fold && s + 1 < nsteps);
        if (rc) return rc;
    }
    return amc_finish_stats(c, sum);
}

int amc_stage_drift(amc_ctx *c, double dt)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    const bool kp = c->keep_prior;
    c->keep_prior = true;       // a following amc_stage_walls needs prior_*_vals
    hipError_t e = amc_launch_stream(c, dt, AMC_ST_DRIFT, 0);
    c->keep_prior = kp;
    AMC_HIP(c, e);
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    return AMC_OK;
}

int amc_stage_walls(amc_ctx *c, amc_step_stats *out)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    AMC_HIP(c, amc_launch_stream(c, 0.0, AMC_ST_WALLS, 0));
    return amc_finish_stats(c, out);
}

int amc_stage_bounds(amc_ctx *c, int64_t *n_moved)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    AMC_HIP(c, amc_launch_stream(c, 0.0, AMC_ST_BOUNDS, 0));
    amc_step_stats st;
    int rc = amc_finish_stats(c, &st);
    if (n_moved) *n_moved = st.n_oob_walls;
    return rc;
}

int amc_stage_sweep(amc_ctx *c, amc_step_stats *out)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    int rc = amc_enqueue_sweep(c);
    if (rc) return rc;
    return amc_finish_stats(c, out);
}

// ---- outputs ----------------------------------------------------------------------------------------------------------
int amc_paths_pending(amc_ctx *c, size_t *n)
{
    if (!c || !n) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    amc_dev_counters now;
    int rc = amc_read_counters(c, &now);
    if (rc) return rc;
    *n = std::min<size_t>(now.path_count, c->out.cap);
    return AMC_OK;
}

int amc_drain_paths(amc_ctx *c, amc_path_record *out, size_t cap, size_t *n)
{
    if (!c || !n) return AMC_ERR_INVALID;
    size_t pending;
    int rc = amc_paths_pending(c, &pending);
    if (rc) return rc;
    if (pending > cap) return amc_fail(c, AMC_ERR_CAPACITY, "amc_drain_paths: %zu records pending, buffer holds %zu", pending, cap);
    if (pending) AMC_HIP(c, hipMemcpyAsync(out, c->d_rec, sizeof(amc_path_record) * pending, hipMemcpyDeviceToHost, c->stream));
    unsigned int zero = 0;
    AMC_HIP(c, hipMemcpyAsync(&c->d_cnt->path_count, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    *n = pending;
    return AMC_OK;
}

int amc_histograms(amc_ctx *c, uint64_t *counts, uint64_t *n_paths_total)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    if (counts) {
        if (!c->d_hist) return amc_fail(c, AMC_ERR_STATE, "histograms disabled (hist_bins == 0)");
    }
    std::vector<uint64_t> banks;
    if (counts) {
        banks.resize((size_t)4 * c->out.nbins * AMC_COUNTER_BANKS);
        AMC_HIP(c, hipMemcpyAsync(banks.data(), c->d_hist, sizeof(uint64_t) * banks.size(), hipMemcpyDeviceToHost, c->stream));
    }
    amc_dev_counters now;
    int rc = amc_read_counters(c, &now);        // (synchronises the stream)
    if (rc) return rc;
    if (counts) {
        const size_t m = (size_t)4 * c->out.nbins;
        for (size_t k = 0; k < m; k++) counts[k] = 0;
        for (int b = 0; b < AMC_COUNTER_BANKS; b++)
            for (size_t k = 0; k < m; k++) counts[k] += banks[(size_t)b * m + k];
    }
    if (n_paths_total) *n_paths_total = now.n_paths_total;
    return AMC_OK;
}

int amc_reset_outputs(amc_ctx *c)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    if (c->d_hist) AMC_HIP(c, hipMemsetAsync(c->d_hist, 0, sizeof(uint64_t) * 4 * c->out.nbins * AMC_COUNTER_BANKS, c->stream));
    AMC_HIP(c, hipMemsetAsync(c->d_cnt, 0, sizeof(amc_dev_counters), c->stream));
    AMC_HIP(c, hipMemsetAsync(c->d_banks, 0, sizeof(amc_counter_bank) * AMC_COUNTER_BANKS, c->stream));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    memset(&c->h_prev, 0, sizeof c->h_prev);
    c->out.step = 0;
    return AMC_OK;
}

// ---- pairwise_particles_in_cell replacement ---------------------------------------------------------------------------
int amc_pairwise_cell(amc_ctx *c, int64_t n_cell, double *continue_path, double *continue_x_path, double *continue_y_path,
                      double *continue_z_path, uint8_t *has_collided, double *x, double *y, double *z, double *vx,
                      double *vy, double *vz, double *out_paths, size_t cap, size_t *n_paths, int64_t *n_collisions)
{
    if (!c) return AMC_ERR_INVALID;
    if (c->P.geometry != AMC_GEOM_CELL) return amc_fail(c, AMC_ERR_STATE, "amc_pairwise_cell needs an AMC_GEOM_CELL context");
    if (n_cell < 0 || n_cell > c->n) return amc_fail(c, AMC_ERR_INVALID, "n_cell %lld exceeds the context capacity %lld", (long long)n_cell, (long long)c->n);
    AMC_HIP(c, hipSetDevice(c->device));
    // the context is sized for its capacity; run on the first n_cell entries
    const int64_t n_full = c->n;
    c->n = n_cell; c->lo = 0; c->hi = n_cell;
    int rc = amc_upload(c, x, y, z, vx, vy, vz, continue_path, continue_x_path, continue_y_path, continue_z_path, has_collided);
    amc_step_stats st;
    memset(&st, 0, sizeof st);
    size_t pending = 0;
    if (!rc) {
        size_t dummy;
        rc = amc_paths_pending(c, &dummy);          // discard nothing: records of earlier calls were drained by them
    }
    if (!rc) rc = amc_timestep(c, 0.0, &st);
    if (!rc) rc = amc_download(c, x, y, z, vx, vy, vz, continue_path, continue_x_path, continue_y_path, continue_z_path, has_collided);
    if (!rc) rc = amc_paths_pending(c, &pending);
    if (!rc && pending) {
        std::vector<amc_path_record> rec(pending);
        size_t got = 0;
        rc = amc_drain_paths(c, rec.data(), pending, &got);
        if (!rc) {
            // the reference appends in loop order: i ascending, j ascending, particle j before particle i
            std::sort(rec.begin(), rec.begin() + got, [](const amc_path_record &a, const amc_path_record &b) {
                if (a.step != b.step) return a.step < b.step;
                if (a.i != b.i) return a.i < b.i;
                if (a.j != b.j) return a.j < b.j;
                return a.which < b.which;
            });
            if (got > cap) rc = amc_fail(c, AMC_ERR_CAPACITY, "out_paths holds %zu paths, %zu were completed", cap, got);
            else if (out_paths)
                for (size_t k = 0; k < got; k++) {
                    out_paths[0 * cap + k] = rec[k].total; out_paths[1 * cap + k] = rec[k].px;
                    out_paths[2 * cap + k] = rec[k].py; out_paths[3 * cap + k] = rec[k].pz;
                }
            pending = got;
        }
    }
    if (n_paths) *n_paths = pending;
    if (n_collisions) *n_collisions = st.n_pp;
    c->n = n_full; c->lo = 0; c->hi = n_full;
    return rc;
}

// ---- measurement ------------------------------------------------------------------------------------------------------
int amc_profile(amc_ctx *c, int enable)
{
    if (!c) return AMC_ERR_INVALID;
    hipSetDevice(c->device);
    amc_prof_collect(c);
    c->profiling = enable != 0;
    return AMC_OK;
}

int amc_overlap_stats(amc_ctx *c, int64_t *out)
{
    if (!c || !out) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    amc_dev_counters now;
    int rc = amc_read_counters(c, &now);
    if (rc) return rc;
    out[0] = c->ovl_steps; out[1] = now.n_refiled; out[2] = c->overlap_mode; out[3] = c->max_extra;
    return AMC_OK;
}

int amc_kernel_times(amc_ctx *c, double *total_ms, int64_t *launches)
{
    if (!c) return AMC_ERR_INVALID;
    hipSetDevice(c->device);
    amc_prof_collect(c);
    if (c->d_dbg) {
        long long h[128];
        hipMemcpy(h, c->d_dbg, sizeof h, hipMemcpyDeviceToHost);
        const double n = h[11] > 0 ? (double)h[11] : 1.0;
        fprintf(stderr, "[amc k_resolve phases, us/launch] count-left %.1f claim %.1f | rounds: collect %.1f pairs %.1f clusters>=3: sort+load %.1f emulate %.1f | overlay %.1f validate %.1f commit %.1f | rounds %.2f cand %.1f complex-members %.2f launches %lld\n",
                h[0] / n / 100.0, h[1] / n / 100.0, h[6] / n / 100.0, h[7] / n / 100.0, h[2] / n / 100.0, h[3] / n / 100.0, h[14] / n / 100.0, h[4] / n / 100.0, h[5] / n / 100.0,
                h[8] / n, h[9] / n, h[10] / n, h[11]);
        {
            // the wide kernel's waves keep their figures in 64 words each (amc_clusters.hip)
            std::vector<long long> wv((size_t)128 * 512);
            hipMemcpy(wv.data(), c->d_dbg + 128, sizeof(long long) * wv.size(), hipMemcpyDeviceToHost);
            long long s[128] = {0}, longest = 0;
            for (int w = 0; w < 512; w++) {
                for (int e = 3; e < 128; e++) if (e != 4) s[e] += wv[(size_t)128 * w + e];
                longest = std::max(longest, wv[(size_t)128 * w + 4]);
            }
            const double nl = h[25] > 0 ? (double)h[25] : 1.0;
            static const char *cn[8] = {"pair", "3-cluster", "4+-cluster", "not owner", "pair+again", "3-cluster+again", "4+-cluster+again", "not owner+again"};
            static const char *pn[12] = {"graph", "walk", "reserve", "small: after emulate", "pair emulate | by the wave: emulate", "publish", "grid probe", "first hop", "particles", "set-up", "overlay probe", "small: prepare"};
            fprintf(stderr, "[amc k_clusters_wide] working waves %lld in %lld launches; launch span (first working wave in -> last out) %.2f us, last out -> ordered workgroup in %.2f us, longest wave ever %.2f us\n",
                    s[3], h[25], h[27] / nl / 100.0, h[26] / nl / 100.0, longest / 100.0);
            fprintf(stderr, "[three-particle path of lane 0] %lld times, %lld continued from the pair's hit; loads + slots %.2f us, emulation %.2f us\n", s[5], s[6],
                    s[7] / (double)std::max<long long>(1, s[5]) / 100.0, s[124] / (double)std::max<long long>(1, s[5]) / 100.0);
            fprintf(stderr, "[pair-wave lifetimes, 2.5 us buckets]");
            for (int k = 0; k < 8; k++) fprintf(stderr, " %lld", s[24 + k]);
            fprintf(stderr, "\n");
            for (int k = 0; k < 8; k++) {
                if (!s[16 + k]) continue;
                fprintf(stderr, "[amc k_clusters_wide %-16s %7lld waves, %5.1f us]", cn[k], s[16 + k], s[8 + k] / (double)s[16 + k] / 100.0);
                for (int e = 0; e < 12; e++) fprintf(stderr, " %s %.2f", pn[e], s[32 + 12 * k + e] / (double)s[16 + k] / 100.0);
                fprintf(stderr, "\n");
            }
        }
    }
    for (int k = 0; k < AMC_K_COUNT; k++) {
        if (total_ms) total_ms[k] = c->k_ms[k];
        if (launches) launches[k] = c->k_launches[k];
    }
    return AMC_OK;
}

}  // extern "C"
