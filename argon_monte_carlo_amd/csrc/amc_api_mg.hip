// amc_api_mg.hip — C ABI of the multi-GPU path (one process per GPU, index-range shards): shard-local step,
// packed position exchange, detection on the whole system, candidate-state exchange and the host-driven resolve rounds.
#include "amc_host.h"

extern "C" {

int amc_set_shard(amc_ctx *c, int64_t lo, int64_t hi)
{
    if (!c || lo < 0 || hi < lo || hi > c->n) return AMC_ERR_INVALID;
    c->lo = lo; c->hi = hi;
    c->mg_count_pp = (lo == 0);     // the rank that owns particle 0 reports the sweep's collision count
    return AMC_OK;
}

static int mg_ensure_xchg(amc_ctx *c)
{
    if (c->xchg_send) return AMC_OK;
    c->xchg_stride = std::max<int64_t>(c->W.max_slots, 1024);
    AMC_HIP(c, hipMalloc(&c->xchg_send, sizeof(double) * 11 * (size_t)c->xchg_stride));
    AMC_HIP(c, hipMalloc(&c->xchg_recv, sizeof(int) * (size_t)c->xchg_stride));
    return AMC_OK;
}

int amc_device_view_get(amc_ctx *c, amc_device_view *out)
{
    if (!c || !out) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = mg_ensure_xchg(c);
    if (rc) return rc;
    out->x = c->S.x; out->y = c->S.y; out->z = c->S.z;
    out->xchg = c->xchg_send; out->xchg_capacity = c->xchg_stride;
    out->n = c->n; out->lo = c->lo; out->hi = c->hi;
    return AMC_OK;
}

int amc_mg_local(amc_ctx *c, double dt)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    { int rc_ = amc_flush(c); if (rc_) return rc_; }
    if (c->allpairs || c->P.geometry == AMC_GEOM_CELL || c->P.geometry == AMC_GEOM_PORE_ENERGISED)
        return amc_fail(c, AMC_ERR_INVALID, "multi-GPU needs the binned detector and the cube / specular pore geometry");
    AMC_HIP(c, hipSetDevice(c->device));
    const int st = (c->P.geometry == AMC_GEOM_CUBE) ? (AMC_ST_DRIFT | AMC_ST_WALLS) : (AMC_ST_DRIFT | AMC_ST_WALLS | AMC_ST_BOUNDS);
    AMC_HIP(c, amc_launch_stream(c, dt, st, 0));
    return AMC_OK;
}

int amc_mg_positions_view(amc_ctx *c, int world, void **send, void **recv, int64_t *m)
{
    if (!c || world < 1 || !send || !recv || !m) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    if (c->pos_world != world) {
        { void *td[] = {c->TD.idx, c->TD.count, c->TD.t, c->TD.contact, c->TD.normal, c->TD.dir, c->TD.Es, c->TD.dpz, c->TD.dE, c->TD.ok};
      for (void *q : td) if (q) hipFree(q); }
    if (c->pos_send) hipFree(c->pos_send);
        if (c->pos_recv) hipFree(c->pos_recv);
        c->pos_send = c->pos_recv = nullptr;
        c->pos_m = (c->n + world - 1) / world;
        const size_t mm = (size_t)std::max<int64_t>(c->pos_m, 1);
        AMC_HIP(c, hipMalloc((void **)&c->pos_send, sizeof(double) * 3 * mm));
        AMC_HIP(c, hipMalloc((void **)&c->pos_recv, sizeof(double) * 3 * mm * (size_t)world));
        c->pos_world = world;
    }
    *send = c->pos_send; *recv = c->pos_recv; *m = c->pos_m;
    return AMC_OK;
}

int amc_mg_pack_positions(amc_ctx *c, int world)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (world != c->pos_world) return amc_fail(c, AMC_ERR_STATE, "amc_mg_positions_view(world=%d) has not been called", world);
    AMC_HIP(c, hipSetDevice(c->device));
    // this rank's range must be the driver's shard of that world size (the unpack side recomputes the ranges)
    AMC_HIP(c, amc_launch_pos_pack(c, world, 0, 0));
    return AMC_OK;
}

int amc_mg_unpack_positions(amc_ctx *c, int world, int rank)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (world != c->pos_world || rank < 0 || rank >= world) return amc_fail(c, AMC_ERR_STATE, "amc_mg_unpack_positions: world/rank do not match amc_mg_positions_view");
    const int64_t base = c->n / world, rem = c->n % world;
    const int64_t lo = rank * base + std::min<int64_t>(rank, rem), hi = lo + base + (rank < rem ? 1 : 0);
    if (lo != c->lo || hi != c->hi) return amc_fail(c, AMC_ERR_STATE, "rank %d of %d owns [%lld,%lld), amc_set_shard says [%lld,%lld)", rank, world, (long long)lo, (long long)hi, (long long)c->lo, (long long)c->hi);
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, amc_launch_pos_pack(c, world, rank, 1));
    return AMC_OK;
}

int amc_mg_detect(amc_ctx *c, int64_t *n_candidates)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, amc_launch_bin(c));
    AMC_HIP(c, amc_launch_detect(c));
    // the counters and the head of the candidate list in one synchronisation (amc_mg_candidates then needs none)
    amc_dev_counters now;
    c->mg_prefix = 0;
    const int pre = std::min(c->W.max_cand, 8192);
    if (c->h_pin && c->h_pin_bytes >= 4096 + 2 * sizeof(int) * (size_t)pre) {
        AMC_HIP(c, hipMemcpyAsync(c->h_pin, c->d_cnt, sizeof now, hipMemcpyDeviceToHost, c->stream));
        AMC_HIP(c, hipMemcpyAsync(c->h_pin + 4096, c->W.cand_i, sizeof(int) * (size_t)pre, hipMemcpyDeviceToHost, c->stream));
        AMC_HIP(c, hipMemcpyAsync(c->h_pin + 4096 + sizeof(int) * (size_t)pre, c->W.cand_j, sizeof(int) * (size_t)pre, hipMemcpyDeviceToHost, c->stream));
        AMC_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(&now, c->h_pin, sizeof now);
        c->mg_prefix = pre;
    } else {
        int rc = amc_read_counters(c, &now);
        if (rc) return rc;
    }
    if (now.cand_count > (unsigned)c->W.max_cand) return amc_fail(c, AMC_ERR_CAPACITY, "candidate list overflow (%u)", now.cand_count);
    if (n_candidates) *n_candidates = now.cand_count;
    c->mg_ncand = (int)now.cand_count;
    return AMC_OK;
}

int amc_mg_candidates(amc_ctx *c, int32_t *cand_i, int32_t *cand_j, size_t cap, size_t *n)
{
    if (!c || !n) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    const size_t k = std::min<size_t>((size_t)std::max(c->mg_ncand, 0), (size_t)c->W.max_cand);   // from amc_mg_detect
    if (k > cap) return amc_fail(c, AMC_ERR_CAPACITY, "amc_mg_candidates: %zu pairs, buffer holds %zu", k, cap);
    if (k && (int)k <= c->mg_prefix) {          // staged by amc_mg_detect
        memcpy(cand_i, c->h_pin + 4096, sizeof(int) * k);
        memcpy(cand_j, c->h_pin + 4096 + sizeof(int) * (size_t)c->mg_prefix, sizeof(int) * k);
    } else if (k) {
        amc_stage stg(c);
        AMC_HIP(c, stg.get(cand_i, c->W.cand_i, sizeof(int) * k));
        AMC_HIP(c, stg.get(cand_j, c->W.cand_j, sizeof(int) * k));
        AMC_HIP(c, stg.finish());
    }
    c->mg_prefix = 0;
    *n = k;
    return AMC_OK;
}

static int mg_upload_list(amc_ctx *c, const int32_t *particles, size_t n)
{
    int rc = mg_ensure_xchg(c);
    if (rc) return rc;
    if ((int64_t)n > c->xchg_stride) return amc_fail(c, AMC_ERR_CAPACITY, "exchange list of %zu particles exceeds capacity %lld", n, (long long)c->xchg_stride);
    if (n) AMC_HIP(c, hipMemcpyAsync(c->xchg_recv, particles, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    return AMC_OK;
}

int amc_mg_pack_state(amc_ctx *c, const int32_t *particles, size_t n)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = mg_upload_list(c, particles, n);
    if (rc) return rc;
    AMC_HIP(c, amc_launch_pack(c, (const int *)c->xchg_recv, (int)n, (double *)c->xchg_send, 0));
    return AMC_OK;
}

int amc_mg_unpack_state(amc_ctx *c, const int32_t *particles, size_t n)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = mg_upload_list(c, particles, n);
    if (rc) return rc;
    AMC_HIP(c, amc_launch_pack(c, (const int *)c->xchg_recv, (int)n, (double *)c->xchg_send, 1));
    return AMC_OK;
}

int amc_mg_exchange_begin(amc_ctx *c, const int32_t *particles, size_t n, size_t *n_rows)
{
    if (!c || !n_rows) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    std::vector<int32_t> list;
    if (!particles) {
        const size_t k = std::min<size_t>((size_t)std::max(c->mg_ncand, 0), (size_t)c->W.max_cand);
        list.resize(2 * k);
        if (k) {
            int rc = amc_mg_candidates(c, list.data(), list.data() + k, k, &n);       // staged by amc_mg_detect: no device access
            if (rc) return rc;
        }
        std::sort(list.begin(), list.end());
        list.erase(std::unique(list.begin(), list.end()), list.end());                // canonical order: ascending particle index
        particles = list.data();
        n = list.size();
    }
    int rc = mg_upload_list(c, particles, n);
    if (rc) return rc;
    AMC_HIP(c, amc_launch_pack(c, (const int *)c->xchg_recv, (int)n, (double *)c->xchg_send, 0));
    c->mg_list_n = n;
    *n_rows = n;
    return AMC_OK;
}

int amc_mg_exchange_end(amc_ctx *c)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, amc_launch_pack(c, (const int *)c->xchg_recv, (int)c->mg_list_n, (double *)c->xchg_send, 1));
    return AMC_OK;
}

int amc_mg_resolve_round(amc_ctx *c, int first, int *dirty, int32_t *new_members, size_t cap, size_t *n_new)
{
    if (!c || !dirty || !n_new) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, amc_launch_resolve_round(c, first));
    amc_resolve_ctl ctl;
    amc_stage stg(c);
    AMC_HIP(c, stg.get(&ctl, c->W.ctl, sizeof ctl));
    AMC_HIP(c, stg.finish());
    *dirty = 0; *n_new = 0;
    if (!ctl.active) return AMC_OK;
    if (ctl.ovf) return amc_fail(c, AMC_ERR_CAPACITY, "resolve work space overflow");
    *dirty = ctl.dirty != 0;
    const int k = ctl.nslots - ctl.nslots0;
    if (k > 0) {
        if ((size_t)k > cap) return amc_fail(c, AMC_ERR_CAPACITY, "%d new cluster members, buffer holds %zu", k, cap);
        AMC_HIP(c, stg.get(new_members, c->W.sl_p + ctl.nslots0, sizeof(int) * (size_t)k));
        AMC_HIP(c, stg.finish());
        std::sort(new_members, new_members + k);
        *n_new = (size_t)k;
    }
    return AMC_OK;
}

int amc_mg_commit(amc_ctx *c)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    AMC_HIP(c, amc_launch_commit(c));
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
