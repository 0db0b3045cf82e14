// amc_api_temp.hip — C ABI of the energised-wall path (Temperature_Pore_MC.py, Temp:662-853): the per-case hand-over to
// the host's random draws (amc_wall_hits / amc_wall_apply) and the opt-in device-side sampling (amc_temp_cases_device).
#include "amc_host.h"

extern "C" {

// ---- energised walls (Temp) ---------------------------------------------------------------------------------------------
static int temp_ensure(amc_ctx *c)
{
    if (c->P.geometry != AMC_GEOM_PORE_ENERGISED) return amc_fail(c, AMC_ERR_STATE, "energised-wall calls need AMC_GEOM_PORE_ENERGISED");
    if (c->T.idx) return AMC_OK;
    amc_temp_ws &T = c->T;
    T.cap = (int)std::min<int64_t>(std::max<int64_t>(4096, c->n / 8 + 1024), 0x3fffffff);
    const size_t cap = (size_t)T.cap;
    // one pinned, device-mapped block: [count | idx | t | contact | normal | dir | Es | dpz | dE | ok]
    size_t off = 0;
    auto place = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
    const size_t o_count = place(64), o_idx = place(sizeof(int) * cap), o_t = place(sizeof(double) * cap),
                 o_contact = place(sizeof(double) * 3 * cap), o_normal = place(sizeof(double) * 3 * cap),
                 o_dir = place(sizeof(double) * 3 * cap), o_Es = place(sizeof(double) * cap), o_dpz = place(sizeof(double) * cap),
                 o_dE = place(sizeof(double) * cap), o_ok = place(cap), o_dEs = place(sizeof(double) * cap),
                 o_ddpz = place(sizeof(double) * cap), o_ddE = place(sizeof(double) * cap);
    void *hp = nullptr, *dp = nullptr;
    AMC_HIP(c, hipHostMalloc(&hp, off, hipHostMallocMapped));
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { hipHostFree(hp); return amc_fail(c, AMC_ERR_HIP, "hipHostGetDevicePointer failed"); }
    memset(hp, 0, off);
    T.pin = hp;
    char *h = (char *)hp, *d = (char *)dp;
    AMC_HIP(c, dalloc(&T.count, 16));       // (the hit counter stays in device memory: every hit increments it atomically)
    T.idx = (int *)(d + o_idx); T.t = (double *)(d + o_t); T.contact = (double *)(d + o_contact);
    T.normal = (double *)(d + o_normal); T.dir = (double *)(d + o_dir); T.Es = (double *)(d + o_Es); T.dpz = (double *)(d + o_dpz);
    T.dE = (double *)(d + o_dE); T.ok = (unsigned char *)(d + o_ok);
    T.h_count = (int *)(h + o_count); T.h_idx = (int *)(h + o_idx); T.h_contact = (double *)(h + o_contact);
    T.h_normal = (double *)(h + o_normal); T.h_dir = (double *)(h + o_dir); T.h_Es = (double *)(h + o_Es);
    T.h_dpz = (double *)(h + o_dpz); T.h_dE = (double *)(h + o_dE);
    T.def_Es = (double *)(d + o_dEs); T.def_dpz = (double *)(d + o_ddpz); T.def_dE = (double *)(d + o_ddE);
    T.h_def_Es = (double *)(h + o_dEs); T.h_def_dpz = (double *)(h + o_ddpz); T.h_def_dE = (double *)(h + o_ddE);
    AMC_HIP(c, dalloc(&T.def_idx, cap));
    AMC_HIP(c, dalloc(&T.def_dir, 3 * cap));
    T.def_case = -1; T.def_n = 0;
    T.last_case = -1; T.last_n = 0; T.pre_case = -1;
    return AMC_OK;
}

int amc_temp_begin(amc_ctx *c, double dt)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = temp_ensure(c);
    if (rc) return rc;
    if ((rc = amc_flush(c))) return rc;
    c->keep_prior = true;       // the energised masks read prior_*_vals (Temp:708-750)
    c->T.pre_case = -1;
    AMC_HIP(c, amc_launch_stream(c, dt, AMC_ST_DRIFT | AMC_ST_WALLS, 0));
    return AMC_OK;
}

int amc_wall_hits(amc_ctx *c, int case_id, int32_t *idx, double *normal_xyz, double *contact_z, size_t cap, size_t *n)
{
    if (!c || !n || case_id < 3 || case_id > 9) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = temp_ensure(c);
    if (rc) return rc;
    amc_temp_ws &T = c->T;
    if (T.pre_case == case_id) {
        T.pre_case = -1;        // launched behind the previous case's apply kernel and already synchronised with it
    } else {
        T.pre_case = -1;
        AMC_HIP(c, amc_launch_temp_hits(c, case_id));
        AMC_HIP(c, hipMemcpyAsync(T.h_count, T.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        AMC_HIP(c, hipStreamSynchronize(c->stream));    // the records are in host memory now (the kernel wrote them there)
    }
    const int cnt = *T.h_count;
    if (cnt > T.cap) return amc_fail(c, AMC_ERR_CAPACITY, "%d wall hits exceed the record capacity %d", cnt, T.cap);
    if ((size_t)cnt > cap) return amc_fail(c, AMC_ERR_CAPACITY, "%d wall hits, caller buffer holds %zu", cnt, cap);
    T.last_case = case_id; T.last_n = cnt;
    T.perm.resize((size_t)cnt);
    *n = (size_t)cnt;
    if (cnt == 0) return AMC_OK;
    const int *hidx = T.h_idx;
    const double *hnorm = T.h_normal, *hcontact = T.h_contact;
    for (int k = 0; k < cnt; k++) T.perm[k] = k;
    std::sort(T.perm.begin(), T.perm.end(), [&](int a, int b) { return hidx[a] < hidx[b]; });   // ascending particle index
    for (int s = 0; s < cnt; s++) {
        const int k = T.perm[s];
        if (idx) idx[s] = hidx[k];
        if (normal_xyz) { normal_xyz[3 * s] = hnorm[3 * k]; normal_xyz[3 * s + 1] = hnorm[3 * k + 1]; normal_xyz[3 * s + 2] = hnorm[3 * k + 2]; }
        if (contact_z) contact_z[s] = hcontact[3 * k + 2];
    }
    return AMC_OK;
}

int amc_wall_apply(amc_ctx *c, int case_id, const double *dir_xyz, const double *surface_energy, size_t n, double *dpz, double *dE)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    amc_temp_ws &T = c->T;
    if (!T.idx || T.last_case != case_id || (size_t)T.last_n != n)
        return amc_fail(c, AMC_ERR_STATE, "amc_wall_apply(case %d, n=%zu) does not match the pending amc_wall_hits(case %d, n=%d)", case_id, n, T.last_case, T.last_n);
    T.last_case = -1;
    if (n == 0) return AMC_OK;
    if (!dir_xyz || !surface_energy) return AMC_ERR_INVALID;
    // caller order (ascending particle index) -> record order, written where the kernel reads it
    for (size_t s = 0; s < n; s++) {
        const int k = T.perm[s];
        T.h_dir[3 * k] = dir_xyz[3 * s]; T.h_dir[3 * k + 1] = dir_xyz[3 * s + 1]; T.h_dir[3 * k + 2] = dir_xyz[3 * s + 2];
        T.h_Es[k] = surface_energy[s];
    }
    AMC_HIP(c, amc_launch_temp_apply(c, case_id, (int)n));
    // The next case's mask is evaluated on the state this apply leaves (Temp:708-751: each mask after the previous handler)
    // and needs nothing from the host: its hits kernel goes right behind, so that ONE synchronisation returns this case's
    // results and the next case's hits (the hand-over of a step is synchronisation latency: 12 -> 7 of them).  The hit
    // records are separate from what the apply kernel wrote back (dpz / dE) and from the host's copy of the permutation.
    int pre = -1;
    if (case_id < 9) {
        AMC_HIP(c, amc_launch_temp_hits(c, case_id + 1));
        AMC_HIP(c, hipMemcpyAsync(T.h_count, T.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        pre = case_id + 1;
    }
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    T.pre_case = pre;
    for (size_t s = 0; s < n; s++) {
        if (dpz) dpz[s] = T.h_dpz[T.perm[s]];
        if (dE) dE[s] = T.h_dE[T.perm[s]];
    }
    return AMC_OK;
}

// A case whose surface energies are not there yet (the gap case: mpmath integrals in worker processes, energised.py) can be
// PARKED: amc_wall_park does everything of amc_wall_apply that does not depend on the energy — the completed path, the
// counters, the particle at its contact point — so that the following cases' masks see what they have to see, and
// amc_wall_finish sets the new velocities when the energies have arrived.  Exact as long as no later case hits a parked
// particle before the finish (the driver checks the hit lists and finishes first if one does; amc_wall_hits_again makes
// the library evaluate that case's hits anew, on the finished state).
int amc_wall_park(amc_ctx *c, int case_id, const double *dir_xyz, size_t n)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    amc_temp_ws &T = c->T;
    if (!T.idx || T.last_case != case_id || (size_t)T.last_n != n)
        return amc_fail(c, AMC_ERR_STATE, "amc_wall_park(case %d, n=%zu) does not match the pending amc_wall_hits(case %d, n=%d)", case_id, n, T.last_case, T.last_n);
    if (T.def_case >= 0) return amc_fail(c, AMC_ERR_STATE, "amc_wall_park: case %d is still parked", T.def_case);
    T.last_case = -1;
    T.def_case = case_id; T.def_n = 0;          // (a shard without a hit of its own parks nothing and finishes nothing)
    if (n == 0) return AMC_OK;
    if (!dir_xyz) return AMC_ERR_INVALID;
    for (size_t s = 0; s < n; s++) {
        const int k = T.perm[s];
        T.h_dir[3 * k] = dir_xyz[3 * s]; T.h_dir[3 * k + 1] = dir_xyz[3 * s + 1]; T.h_dir[3 * k + 2] = dir_xyz[3 * s + 2];
        T.h_Es[k] = 0.0;
    }
    AMC_HIP(c, amc_launch_temp_apply(c, case_id, (int)n, true));
    T.def_case = case_id; T.def_n = (int)n; T.def_perm = T.perm;
    int pre = -1;
    if (case_id < 9) {                          // (the next case's hits behind it, as in amc_wall_apply)
        AMC_HIP(c, amc_launch_temp_hits(c, case_id + 1));
        AMC_HIP(c, hipMemcpyAsync(T.h_count, T.count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        pre = case_id + 1;
    }
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    T.pre_case = pre;
    return AMC_OK;
}

int amc_wall_finish(amc_ctx *c, int case_id, const double *surface_energy, size_t n, double *dpz, double *dE)
{
    if (!c) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    amc_temp_ws &T = c->T;
    if (T.def_case != case_id || (size_t)T.def_n != n)
        return amc_fail(c, AMC_ERR_STATE, "amc_wall_finish(case %d, n=%zu) does not match the parked case %d (n=%d)", case_id, n, T.def_case, T.def_n);
    T.def_case = -1;
    if (n == 0) return AMC_OK;
    if (!surface_energy) return AMC_ERR_INVALID;
    for (size_t s = 0; s < n; s++) T.h_def_Es[T.def_perm[s]] = surface_energy[s];
    AMC_HIP(c, amc_launch_temp_velocity(c, case_id, (int)n));
    AMC_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t s = 0; s < n; s++) {
        if (dpz) dpz[s] = T.h_def_dpz[T.def_perm[s]];
        if (dE) dE[s] = T.h_def_dE[T.def_perm[s]];
    }
    return AMC_OK;
}

int amc_wall_hits_again(amc_ctx *c)
{
    if (!c) return AMC_ERR_INVALID;
    c->T.pre_case = -1;             // the hits launched behind the last apply are not used: the next amc_wall_hits evaluates its case anew
    c->T.last_case = -1;
    return AMC_OK;
}

// ---- device-RNG mode ---------------------------------------------------------------------------------------------------------
static int temp_dev_ensure(amc_ctx *c)
{
    if (c->P.geometry != AMC_GEOM_PORE_ENERGISED) return amc_fail(c, AMC_ERR_STATE, "energised-wall calls need AMC_GEOM_PORE_ENERGISED");
    amc_temp_dev_ws &D = c->TD;
    if (D.idx) return AMC_OK;
    D.cap = (int)std::min<int64_t>(std::max<int64_t>(4096, c->n / 64 + 1024), 0x0fffffff);
    const size_t cap = (size_t)D.cap * 7;
    AMC_HIP(c, dalloc(&D.idx, cap)); AMC_HIP(c, dalloc(&D.count, 8)); AMC_HIP(c, dalloc(&D.t, cap));
    AMC_HIP(c, dalloc(&D.contact, 3 * cap)); AMC_HIP(c, dalloc(&D.normal, 3 * cap)); AMC_HIP(c, dalloc(&D.dir, 3 * cap));
    AMC_HIP(c, dalloc(&D.Es, cap)); AMC_HIP(c, dalloc(&D.dpz, cap)); AMC_HIP(c, dalloc(&D.dE, cap)); AMC_HIP(c, dalloc(&D.ok, cap));
    return AMC_OK;
}

int amc_temp_cases_device(amc_ctx *c, const amc_temp_rng *cfg)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    if (!cfg || cfg->struct_size != (int32_t)sizeof(amc_temp_rng) || cfg->n_gl < 2 || cfg->n_gl > 32)
        return amc_fail(c, AMC_ERR_INVALID, "amc_temp_rng: bad struct_size / n_gl");
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = temp_dev_ensure(c);
    if (rc) return rc;
    if (!c->keep_prior) return amc_fail(c, AMC_ERR_STATE, "amc_temp_cases_device follows amc_temp_begin");
    c->TD.fetched = false;
    AMC_HIP(c, amc_launch_temp_cases_device(c, cfg));
    return AMC_OK;
}

// counts of all seven cases and the head of every segment in one synchronisation; longer segments are completed here
static int temp_dev_fetch(amc_ctx *c)
{
    amc_temp_dev_ws &D = c->TD;
    if (D.fetched) return AMC_OK;
    const int pre = std::min(D.cap, 2048);
    {
        amc_stage st(c);
        AMC_HIP(c, st.get(D.h_count, D.count, sizeof(int) * 7));
        for (int s = 0; s < 7; s++) {
            const size_t o = (size_t)s * (size_t)D.cap;
            D.h_idx[s].resize((size_t)pre); D.h_dpz[s].resize((size_t)pre); D.h_dE[s].resize((size_t)pre); D.h_ok[s].resize((size_t)pre);
            AMC_HIP(c, st.get(D.h_idx[s].data(), D.idx + o, sizeof(int) * (size_t)pre));
            AMC_HIP(c, st.get(D.h_dpz[s].data(), D.dpz + o, sizeof(double) * (size_t)pre));
            AMC_HIP(c, st.get(D.h_dE[s].data(), D.dE + o, sizeof(double) * (size_t)pre));
            AMC_HIP(c, st.get(D.h_ok[s].data(), D.ok + o, (size_t)pre));
        }
        AMC_HIP(c, st.finish());
    }
    for (int s = 0; s < 7; s++) {
        if (D.h_count[s] > D.cap) return amc_fail(c, AMC_ERR_CAPACITY, "%d wall hits in case %d exceed the record capacity %d", D.h_count[s], 3 + s, D.cap);
        const size_t k = (size_t)std::max(D.h_count[s], 0);
        if ((int)k > pre) {
            const size_t o = (size_t)s * (size_t)D.cap;
            D.h_idx[s].resize(k); D.h_dpz[s].resize(k); D.h_dE[s].resize(k); D.h_ok[s].resize(k);
            amc_stage st(c);
            AMC_HIP(c, st.get(D.h_idx[s].data(), D.idx + o, sizeof(int) * k));
            AMC_HIP(c, st.get(D.h_dpz[s].data(), D.dpz + o, sizeof(double) * k));
            AMC_HIP(c, st.get(D.h_dE[s].data(), D.dE + o, sizeof(double) * k));
            AMC_HIP(c, st.get(D.h_ok[s].data(), D.ok + o, k));
            AMC_HIP(c, st.finish());
        }
    }
    D.fetched = true;
    return AMC_OK;
}

int amc_temp_device_results(amc_ctx *c, int case_id, int32_t *idx, double *dpz, double *dE, uint8_t *ok, size_t cap, size_t *n)
{
    if (!c || !n || case_id < 3 || case_id > 9 || !c->TD.idx) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = temp_dev_fetch(c);
    if (rc) return rc;
    amc_temp_dev_ws &D = c->TD;
    const int s = case_id - 3;
    const size_t k = (size_t)std::max(D.h_count[s], 0);
    if (k > cap) return amc_fail(c, AMC_ERR_CAPACITY, "%zu wall hits, caller buffer holds %zu", k, cap);
    std::vector<int> perm(k);
    for (size_t u = 0; u < k; u++) perm[u] = (int)u;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return D.h_idx[s][a] < D.h_idx[s][b]; });   // ascending particle index
    for (size_t u = 0; u < k; u++) {
        const int r = perm[u];
        if (idx) idx[u] = D.h_idx[s][r];
        if (dpz) dpz[u] = D.h_dpz[s][r];
        if (dE) dE[u] = D.h_dE[s][r];
        if (ok) ok[u] = D.h_ok[s][r];
    }
    *n = k;
    return AMC_OK;
}

int amc_temp_device_sums(amc_ctx *c, double *sums, int32_t *had)
{
    if (!c || !sums || !had || !c->TD.idx) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = temp_dev_fetch(c);
    if (rc) return rc;
    amc_temp_dev_ws &D = c->TD;
    sums[0] = sums[1] = sums[2] = 0.0;
    had[0] = had[1] = had[2] = 0;
    std::vector<int> perm;
    for (int s = 0; s < 7; s++) {
        const int case_id = 3 + s;
        const size_t k = (size_t)std::max(D.h_count[s], 0);
        if (!k) continue;
        perm.resize(k);
        for (size_t u = 0; u < k; u++) perm[u] = (int)u;
        std::sort(perm.begin(), perm.end(), [&](int a, int b) { return D.h_idx[s][a] < D.h_idx[s][b]; });
        double m_case = 0.0, e_case = 0.0;
        bool any = false;
        for (size_t u = 0; u < k; u++) {
            const int r = perm[u];
            if (!D.h_ok[s][r]) continue;
            m_case = m_case + D.h_dpz[s][r];
            e_case = e_case + D.h_dE[s][r];
            any = true;
        }
        sums[0] = sums[0] + m_case;
        had[0] |= any ? 1 : 0;
        const bool cold = (case_id == 3 || case_id == 7 || case_id == 9), hot = (case_id == 4 || case_id == 6 || case_id == 8);
        if (cold) { sums[1] = sums[1] + e_case; had[1] |= any ? 1 : 0; }
        if (hot) { sums[2] = sums[2] + e_case; had[2] |= any ? 1 : 0; }
    }
    return AMC_OK;
}

int amc_temp_device_draws(amc_ctx *c, int case_id, int32_t *idx, double *normal_xyz, double *contact_z, double *dir_xyz,
                          double *surface_energy, size_t cap, size_t *n)
{
    if (!c || !n || case_id < 3 || case_id > 9 || !c->TD.idx) return AMC_ERR_INVALID;
    AMC_HIP(c, hipSetDevice(c->device));
    int rc = temp_dev_fetch(c);
    if (rc) return rc;
    amc_temp_dev_ws &D = c->TD;
    const int s = case_id - 3;
    const size_t k = (size_t)std::max(D.h_count[s], 0), o = (size_t)s * (size_t)D.cap;
    if (k > cap) return amc_fail(c, AMC_ERR_CAPACITY, "%zu wall hits, caller buffer holds %zu", k, cap);
    *n = k;
    if (!k) return AMC_OK;
    std::vector<double> hn(3 * k), hc(3 * k), hd(3 * k), he(k);
    AMC_HIP(c, hipMemcpy(hn.data(), D.normal + 3 * o, sizeof(double) * 3 * k, hipMemcpyDeviceToHost));
    AMC_HIP(c, hipMemcpy(hc.data(), D.contact + 3 * o, sizeof(double) * 3 * k, hipMemcpyDeviceToHost));
    AMC_HIP(c, hipMemcpy(hd.data(), D.dir + 3 * o, sizeof(double) * 3 * k, hipMemcpyDeviceToHost));
    AMC_HIP(c, hipMemcpy(he.data(), D.Es + o, sizeof(double) * k, hipMemcpyDeviceToHost));
    std::vector<int> perm(k);
    for (size_t u = 0; u < k; u++) perm[u] = (int)u;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return D.h_idx[s][a] < D.h_idx[s][b]; });
    for (size_t u = 0; u < k; u++) {
        const int r = perm[u];
        if (idx) idx[u] = D.h_idx[s][r];
        for (int e = 0; e < 3; e++) {
            if (normal_xyz) normal_xyz[3 * u + e] = hn[3 * r + e];
            if (dir_xyz) dir_xyz[3 * u + e] = hd[3 * r + e];
        }
        if (contact_z) contact_z[u] = hc[3 * r + 2];
        if (surface_energy) surface_energy[u] = he[r];
    }
    return AMC_OK;
}

int amc_temp_end(amc_ctx *c, amc_step_stats *out)
{
    if (!c || !c->uploaded) return AMC_ERR_STATE;
    AMC_HIP(c, hipSetDevice(c->device));
    if (c->P.geometry != AMC_GEOM_PORE_ENERGISED) return amc_fail(c, AMC_ERR_STATE, "amc_temp_end needs AMC_GEOM_PORE_ENERGISED");
    c->T.pre_case = -1;
    // the bounds pass before the sweep sees every particle at its final pre-sweep position: it builds the detection
    // grid's lists as well (like the fused streaming pass of the specular geometries)
    const bool fuse = !c->allpairs && c->lo == 0 && c->hi == c->n;
    AMC_HIP(c, amc_launch_stream(c, 0.0, AMC_ST_BOUNDS, 0, fuse));  // Temp:804
    int rc = amc_enqueue_sweep(c, fuse);                                // Temp:813-842
    if (rc) return rc;
    AMC_HIP(c, amc_launch_stream(c, 0.0, AMC_ST_BOUNDS, 1));        // Temp:844
    c->out.step++;
    return amc_finish_stats(c, out);
}

}  // extern "C"
