// amc_host.h — host-side helpers shared by the C ABI translation units (amc_api*.hip): device allocation, the pinned
// staging of small read-backs, counters / per-step statistics, the deferred commit and the sweep driver.
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "amc_internal.h"

template <class T>
static hipError_t dalloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    return hipMalloc((void **)p, sizeof(T) * count);
}

// Small device -> host read-backs go through the pinned staging buffer: queue any number of pieces, synchronise once,
// then copy out.  (Falls back to direct copies when a piece does not fit.)
struct amc_stage {
    amc_ctx *c;
    size_t off = 0;
    struct piece { void *dst; size_t off, bytes; };
    std::vector<piece> pieces;
    explicit amc_stage(amc_ctx *ctx) : c(ctx) {}
    hipError_t get(void *dst, const void *src, size_t bytes)
    {
        if (!bytes) return hipSuccess;
        const size_t at = (off + 63) & ~(size_t)63;
        if (!c->h_pin || at + bytes > c->h_pin_bytes) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
        pieces.push_back({dst, at, bytes});
        off = at + bytes;
        return hipMemcpyAsync(c->h_pin + at, src, bytes, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t finish()
    {
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess)
            for (auto &p : pieces) memcpy(p.dst, c->h_pin + p.off, p.bytes);
        pieces.clear();
        off = 0;
        return e;
    }
};

// (defined in amc_api.hip inside its extern "C" block; internal to the library, not part of the ABI)
#define AMC_INTERNAL extern "C" __attribute__((visibility("hidden")))
AMC_INTERNAL int amc_read_counters(amc_ctx *c, amc_dev_counters *h);    // device counters with the banks folded in (synchronises)
AMC_INTERNAL int amc_finish_stats(amc_ctx *c, amc_step_stats *out);     // per-step deltas + error flags
AMC_INTERNAL int amc_publish_velocities(amc_ctx *c);                    // multi-GPU: vpub = current velocities (after an upload)
AMC_INTERNAL int amc_flush(amc_ctx *c);                                 // write deferred sweep results to the particle arrays
AMC_INTERNAL int amc_enqueue_sweep(amc_ctx *c, bool counted = false, bool defer_commit = false);   // bin (unless counted) + detect + resolve
