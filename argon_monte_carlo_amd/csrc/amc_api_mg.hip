// amc_api_mg.hip — C ABI of the multi-GPU path (one process per GPU, index-range shards): shard-local step,
// packed exchange of positions and changed velocities, then the single-GPU sweep over the whole system on every rank.
#include "amc_host.h"

extern "C" {

int amc_set_shard(amc_ctx *c, int64_t lo, int64_t hi)
{
    if (!c || lo < 0 || hi < lo || hi > c->n) return AMC_ERR_INVALID;
    c->lo = lo; c->hi = hi;
    c->mg_count_pp = (lo == 0);     // the rank that owns particle 0 reports the sweep's collision count
    AMC_HIP(c, hipSetDevice(c->device));
    if (!c->kin_vpub) AMC_HIP(c, hipMalloc((void **)&c->kin_vpub, sizeof(double) * 3 * (size_t)std::max<int64_t>(c->n, 1)));
    if (c->uploaded) return amc_publish_velocities(c);
    return AMC_OK;
}

int amc_mg_local(amc_ctx *c, double dt)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (c->allpairs || c->P.geometry == AMC_GEOM_CELL || c->P.geometry == AMC_GEOM_PORE_ENERGISED)
        return amc_fail(c, AMC_ERR_INVALID, "amc_mg_local needs the binned detector and the cube / specular pore geometry (energised walls: amc_temp_begin)");
    AMC_HIP(c, hipSetDevice(c->device));
    const int st = (c->P.geometry == AMC_GEOM_CUBE) ? (AMC_ST_DRIFT | AMC_ST_WALLS) : (AMC_ST_DRIFT | AMC_ST_WALLS | AMC_ST_BOUNDS);
    AMC_HIP(c, amc_launch_stream(c, dt, st, 0));
    return AMC_OK;
}

int amc_mg_exchange_view(amc_ctx *c, int world, void **send, void **recv, int64_t *block)
{
    if (!c || world < 1 || !send || !recv || !block) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    if (c->kin_world != world) {
        AMC_HIP(c, hipStreamSynchronize(c->stream));
        if (c->kin_send) hipFree(c->kin_send);
        if (c->kin_recv) hipFree(c->kin_recv);
        c->kin_send = c->kin_recv = nullptr;
        c->kin_world = 0;
        c->kin_m = std::max<int64_t>((c->n + world - 1) / world, 1);
        c->kin_cap = std::max<int64_t>(4096, c->kin_m / 8);
        if (const char *e = getenv("AMC_MG_VELOCITY_LIST")) { const long long v = atoll(e); if (v > 0) c->kin_cap = v; }   // (tests)
        const int64_t kb = amc_kin_banks();
        c->kin_cap = (c->kin_cap + kb - 1) / kb * kb;                   // the same room in every bank
        c->kin_block = 3 * c->kin_m + kb + 4 * c->kin_cap;
        AMC_HIP(c, hipMalloc((void **)&c->kin_send, sizeof(double) * (size_t)c->kin_block));
        AMC_HIP(c, hipMalloc((void **)&c->kin_recv, sizeof(double) * (size_t)c->kin_block * (size_t)world));
        c->kin_world = world;
        c->kin_counts_clear = false;
        // kept lists (pore): a node pool per wave of the pack and of the unpack kernel, whose launch geometry is fixed by the
        // world size; the node records behind the particles' own grow if these pools need more than the single-GPU pass's
        c->mg_keep = false;
        c->lists_age = -1;
        if (c->mg_wave_count) { hipFree(c->mg_wave_count); c->mg_wave_count = nullptr; }
        if (c->keep_K >= 2 && c->B.cell_of && c->B_buf[0].rec == c->B.rec) {
            const int64_t per = std::max<int64_t>(c->kin_m, c->kin_cap);
            c->mg_waves_pack = (int)((c->kin_m + 255) / 256) * 4;
            c->mg_waves_unpack = (int)(((int64_t)world * per + 255) / 256) * 4;
            const size_t waves = (size_t)c->mg_waves_pack + (size_t)c->mg_waves_unpack;
            const size_t pool = waves * (size_t)c->B.wave_cap;
            if ((long long)c->n + (long long)pool <= 0x3fffffffLL) {
                if (pool > c->keep_pool) {
                    amc_rec *rec = nullptr; int *extra = nullptr;
                    AMC_HIP(c, dalloc(&rec, (size_t)c->n + std::max((size_t)c->max_extra, pool)));
                    AMC_HIP(c, dalloc(&extra, pool));
                    hipFree(c->B.rec); hipFree(c->B.extra);
                    c->B.rec = rec; c->B.extra = extra; c->keep_pool = pool;
                    c->B_buf[0].rec = rec; c->B_buf[0].extra = extra;
                }
                AMC_HIP(c, dalloc(&c->mg_wave_count, waves));
                AMC_HIP(c, hipMemsetAsync(c->mg_wave_count, 0, sizeof(int) * waves, c->stream));
                c->mg_keep = true;
            }
        }
    }
    *send = c->kin_send; *recv = c->kin_recv; *block = c->kin_block;
    return AMC_OK;
}

// this rank's range must be the driver's shard of that world size (the unpack side recomputes the ranges)
static int mg_check_shard(amc_ctx *c, int world, int rank)
{
    if (world != c->kin_world || rank < 0 || rank >= world) return amc_fail(c, AMC_ERR_STATE, "world/rank do not match amc_mg_exchange_view");
    const int64_t base = c->n / world, rem = c->n % world;
    const int64_t lo = rank * base + std::min<int64_t>(rank, rem), hi = lo + base + (rank < rem ? 1 : 0);
    if (lo != c->lo || hi != c->hi) return amc_fail(c, AMC_ERR_STATE, "rank %d of %d owns [%lld,%lld), amc_set_shard says [%lld,%lld)", rank, world, (long long)lo, (long long)hi, (long long)c->lo, (long long)c->hi);
    return AMC_OK;
}

int amc_mg_pack(amc_ctx *c, int world)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (world != c->kin_world) return amc_fail(c, AMC_ERR_STATE, "amc_mg_exchange_view(world=%d) has not been called", world);
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    AMC_HIP(c, amc_launch_kin_pack(c, world, 0, 0));
    return AMC_OK;
}

int amc_mg_sweep(amc_ctx *c, int world, int rank)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (c->allpairs || c->P.geometry == AMC_GEOM_CELL) return amc_fail(c, AMC_ERR_INVALID, "multi-GPU needs the binned detector");
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    if (world > 1) {
        int rc = mg_check_shard(c, world, rank);
        if (rc) return rc;
        if (!c->kin_lists) return amc_fail(c, AMC_ERR_STATE, "amc_mg_sweep(world=%d) without amc_mg_pack in this step", world);
        AMC_HIP(c, amc_launch_kin_pack(c, world, rank, 1));
    }
    // the single-GPU sweep over all n particles; the per-cell lists were built by the pack / unpack kernels (if nothing
    // was packed — one rank, no exchange — they are built here).  The scatter of the results is deferred as on one GPU:
    // the next streaming pass over the shard picks up those of its own particles, the next unpack releases the slots of
    // the others (whose results arrive from their owners).
    const bool lists = c->kin_lists;
    c->kin_lists = false;
    return amc_enqueue_sweep(c, lists, true);
}

// ---- detection sharded by index: unpack + detect over [lo, hi) | second all-gather (candidate pairs) | graph + resolve ----
int amc_mg_candidates_view(amc_ctx *c, int world, void **send, void **recv, int64_t *block_ints)
{
    if (!c || world < 1 || !send || !recv || !block_ints) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    if (c->cand_world != world) {
        AMC_HIP(c, hipStreamSynchronize(c->stream));
        if (c->cand_send) hipFree(c->cand_send);
        if (c->cand_recv) hipFree(c->cand_recv);
        c->cand_send = c->cand_recv = nullptr;
        // a quarter of the context's candidate capacity (n / 32 pairs by default; amc_params.max_candidates scales it): ~60x the
        // pairs a rank of eight finds per step at the reference's density, and room for the first step of a synthetic start,
        // whose uniformly placed particles overlap by the ten thousand
        long long cap = std::max<long long>(4096, c->W.max_cand / 4);
        if (const char *e = getenv("AMC_MG_CANDIDATES")) { const long long v = atoll(e); if (v > 0) cap = v; }      // (tests)
        c->cand_cap = (int)std::min<long long>(cap, c->W.max_cand);
        const size_t blk = (size_t)2 + 2 * (size_t)c->cand_cap;
        AMC_HIP(c, dalloc(&c->cand_send, blk));
        AMC_HIP(c, dalloc(&c->cand_recv, blk * (size_t)world));
        AMC_HIP(c, hipMemsetAsync(c->cand_send, 0, sizeof(int) * blk, c->stream));
        AMC_HIP(c, hipMemsetAsync(c->cand_recv, 0, sizeof(int) * blk * (size_t)world, c->stream));
        AMC_HIP(c, hipStreamSynchronize(c->stream));
        c->cand_world = world;
    }
    *send = c->cand_send; *recv = c->cand_recv; *block_ints = 2 + 2 * (int64_t)c->cand_cap;
    return AMC_OK;
}

int amc_mg_detect(amc_ctx *c, int world, int rank)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (c->allpairs || c->detect_ap || c->P.geometry == AMC_GEOM_CELL) return amc_fail(c, AMC_ERR_INVALID, "multi-GPU needs the binned detector");
    if (world != c->cand_world) return amc_fail(c, AMC_ERR_STATE, "amc_mg_candidates_view(world=%d) has not been called", world);
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    int rc = mg_check_shard(c, world, rank);
    if (rc) return rc;
    if (!c->kin_lists) return amc_fail(c, AMC_ERR_STATE, "amc_mg_detect(world=%d) without amc_mg_pack in this step", world);
    if (world > 1) AMC_HIP(c, amc_launch_kin_pack(c, world, rank, 1));          // the other shards' positions in, lists completed
    c->kin_lists = false;
    AMC_HIP(c, amc_launch_detect_own(c));
    return AMC_OK;
}

int amc_mg_resolve(amc_ctx *c, int world)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (world != c->cand_world) return amc_fail(c, AMC_ERR_STATE, "amc_mg_candidates_view(world=%d) has not been called", world);
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, amc_launch_ingest(c, world));
    AMC_HIP(c, amc_launch_resolve(c, true));
    c->lazy_pending = true;
    return AMC_OK;
}

int amc_mg_bounds(amc_ctx *c)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    AMC_HIP(c, amc_launch_stream(c, 0.0, AMC_ST_BOUNDS, 0));        // Temp:804 on the owned range, counters read later
    return AMC_OK;
}

int amc_mg_finish(amc_ctx *c, amc_step_stats *out)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    if (c->P.geometry == AMC_GEOM_PORE || c->P.geometry == AMC_GEOM_PORE_ENERGISED)
        AMC_HIP(c, amc_launch_stream(c, 0.0, AMC_ST_BOUNDS, 1));                     // Pore:550 / Temp:844
    c->out.step++;
    if (!out) return AMC_OK;        // asynchronous: the caller reads the counters later
    return amc_finish_stats(c, out);
}

}  // extern "C"
