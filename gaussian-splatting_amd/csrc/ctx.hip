// Device context: replaces setup_gpu() (setup_gpu.py:3-6,370).
#include "common.h"

static_assert(sizeof(gs_camera) == 144, "gs_camera layout is part of the C ABI (tests/test_abi.py mirrors it)");
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>

static thread_local char g_err[512] = "";

int gs_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *gs_last_error(void) { return g_err; }
extern "C" int gs_version(void) { return 1; }

int gs_buf_reserve(gs_ctx *ctx, gs_buf *b, size_t bytes, bool exact) {
    if (bytes <= b->cap) return GS_OK;
    // the old block may still be in use by work queued on the stream
    GS_HIP(hipStreamSynchronize(ctx->stream));
    if (b->p) GS_HIP(hipFree(b->p));
    ctx->workspace_bytes -= b->cap;
    b->p = nullptr;
    b->cap = 0;
    size_t want = exact ? bytes + 256 : bytes + bytes / 4 + 256;   // reserve-ahead calls name the final size
    GS_HIP(hipMalloc(&b->p, want));
    b->cap = want;
    ctx->workspace_bytes += want;
    ctx->allocations++;      // gs_ctx_info("allocations"): a caller that reserved ahead checks that a step adds none
    return GS_OK;
}

// The library's streams: default priority only.  Round 2 measured a scheduler that put the list kernels on a
// high-priority stream in front of default-priority blend streams, with events between them: a later
// gs_bin_count_end never returned (DESIGN.md, section Concurrency).  The cause was not established beyond that
// arrangement, so it is refused at the boundary instead of being offered.
static int check_stream_priority(gs_ctx *ctx, hipStream_t stream) {
    for (hipStream_t s : ctx->checked_streams)
        if (s == stream) return GS_OK;
    int prio = 0;
    GS_HIP(hipStreamGetPriority(stream, &prio));
    GS_REQUIRE(prio == 0, "stream %p has priority %d: the library launches on default-priority streams only "
               "(priority streams with cross-stream event waits hung the GPU, DESIGN.md section Concurrency)",
               (void *)stream, prio);
    if (ctx->checked_streams.size() < 64) ctx->checked_streams.push_back(stream);
    return GS_OK;
}

// Host wait for `ev` that cannot hang the process: polls until the event has completed or limit_ms have
// passed (GSPLAT_WAIT_TIMEOUT_MS, default 30 s), then reports an error.  The first 200 us spin (a tile count is
// on the host within that), later polls yield the core.
int gs_wait_event(gs_ctx *ctx, hipEvent_t ev, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return GS_OK;
        if (e != hipErrorNotReady)
            return gs_set_error(GS_ERR_HIP, "%s: hipEventQuery failed: %s", what, hipGetErrorString(e));
        if ((spins & 15) == 15) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > ctx->wait_limit_ms)
                return gs_set_error(GS_ERR_HIP, "%s: the device has not finished after %.0f ms (GSPLAT_WAIT_TIMEOUT_MS); the "
                                    "stream is stuck behind work that does not complete", what, ms);
            if (ms > 0.2) std::this_thread::yield();
        }
    }
}

extern "C" int gs_ctx_create(int device, void *stream, gs_ctx **out) {
    GS_REQUIRE(out != nullptr, "gs_ctx_create: out is NULL");
    int count = 0;
    GS_HIP(hipGetDeviceCount(&count));
    GS_REQUIRE(device >= 0 && device < count, "gs_ctx_create: device %d out of range (%d devices)",
               device, count);
    GS_HIP(hipSetDevice(device));
    {   // once per device, with that device current: the address of its copy of the static scratch array
        const int rc = gs_scratch_setup(device);
        if (rc) return rc;
    }
    gs_ctx *c = new gs_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    if (const int rc = check_stream_priority(c, c->stream)) {
        delete c;
        return rc;
    }
    if (const char *e = getenv("GSPLAT_WAIT_TIMEOUT_MS")) {
        const double v = atof(e);
        if (v > 0) c->wait_limit_ms = v;
    }
    hipError_t e = hipHostMalloc((void **)&c->total_host, (2 + 2 * GS_BIN_MAX_VIEWS) * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        delete c;
        return gs_set_error(GS_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    if (const char *e = getenv("GSPLAT_TILE_ORDER")) c->tile_order = atoi(e) != 0;
    if (const char *e = getenv("GSPLAT_SORTED_COUNT_FROM")) c->sorted_count_from = atoi(e) >= 0 ? atoi(e) : c->sorted_count_from;
    if (const char *e = getenv("GSPLAT_PPT")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) c->ppt = v;
    }
    *out = c;
    return GS_OK;
}

extern "C" int gs_ctx_set_option(gs_ctx *ctx, const char *name, int32_t value) {
    GS_REQUIRE(ctx && name, "gs_ctx_set_option: ctx/name is NULL");
    if (strcmp(name, "pixels_per_thread") == 0) {
        GS_REQUIRE(value == 1 || value == 2 || value == 4, "pixels_per_thread must be 1, 2 or 4 (got %d)", value);
        ctx->ppt = value;
        return GS_OK;
    }
    if (strcmp(name, "deterministic_backward") == 0) {
        GS_REQUIRE(value == 0 || value == 1, "deterministic_backward must be 0 or 1 (got %d)", value);
        ctx->deterministic_bwd = value != 0;
        return GS_OK;
    }
    if (strcmp(name, "tile_order") == 0) {
        GS_REQUIRE(value == 0 || value == 1, "tile_order must be 0 or 1 (got %d)", value);
        ctx->tile_order = value != 0;
        return GS_OK;
    }
    if (strcmp(name, "tile_histogram") == 0) {
        GS_REQUIRE(value == 0 || value == 1, "tile_histogram must be 0 or 1 (got %d)", value);
        ctx->tile_hist = value != 0;
        return GS_OK;
    }
    if (strcmp(name, "cull_shrink_rect") == 0) {
        GS_REQUIRE(value == 0 || value == 1, "cull_shrink_rect must be 0 or 1 (got %d)", value);
        ctx->cull_shrink = value != 0;
        return GS_OK;
    }
    if (strcmp(name, "sorted_count_from") == 0) {   // gaussians from which the count runs in depth order (0: always)
        GS_REQUIRE(value >= 0, "sorted_count_from must be >= 0 (got %d)", value);
        ctx->sorted_count_from = value;
        return GS_OK;
    }
    if (strcmp(name, "profile_mask") == 0) {   // bit i: time kernel id i (gs_profile_kernel_name) when profiling is on
        ctx->prof_mask = (uint32_t)value;
        return GS_OK;
    }
    return gs_set_error(GS_ERR_ARG, "gs_ctx_set_option: unknown option '%s'", name);
}

static const char *const k_kernel_names[GS_K_NUM] = {
    "preprocess", "bin_count", "bin_scan", "bin_offsets", "bin_duplicate", "bin_sort", "bin_ranges",
    "blend_forward", "blend_backward", "grad_unpack", "param_backward", "blend_pack", "bin_depth_sort", "image_loss", "adam_step"};

static hipEvent_t prof_event(gs_ctx *c) {
    if (!c->prof_free.empty()) {
        hipEvent_t e = c->prof_free.back();
        c->prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

gs_prof_scope::gs_prof_scope(gs_ctx *c, int kernel) : ctx(c) {
    if (!c || !c->prof_on || !((c->prof_mask >> kernel) & 1u)) return;
    span.kernel = kernel;
    span.start = prof_event(c);
    span.stop = prof_event(c);
    if (!span.start || !span.stop) return;
    live = hipEventRecord(span.start, c->stream) == hipSuccess;
}

gs_prof_scope::~gs_prof_scope() {
    if (!live) return;
    if (hipEventRecord(span.stop, ctx->stream) == hipSuccess) ctx->prof_spans.push_back(span);
}

extern "C" int gs_profile_kernel_count(void) { return GS_K_NUM; }
extern "C" const char *gs_profile_kernel_name(int i) {
    return (i >= 0 && i < GS_K_NUM) ? k_kernel_names[i] : nullptr;
}

extern "C" int gs_ctx_profile(gs_ctx *ctx, int32_t enable) {
    GS_REQUIRE(ctx, "gs_ctx_profile: ctx is NULL");
    ctx->prof_on = enable != 0;
    return GS_OK;
}

extern "C" int gs_ctx_profile_read(gs_ctx *ctx, double *ms_sum, int64_t *launches) {
    GS_REQUIRE(ctx && ms_sum && launches, "gs_ctx_profile_read: NULL argument");
    for (int i = 0; i < GS_K_NUM; i++) { ms_sum[i] = 0.0; launches[i] = 0; }
    GS_HIP(hipSetDevice(ctx->device));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    for (gs_prof_span &s : ctx->prof_spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
            ms_sum[s.kernel] += ms;
            launches[s.kernel] += 1;
        }
        ctx->prof_free.push_back(s.start);
        ctx->prof_free.push_back(s.stop);
    }
    ctx->prof_spans.clear();
    return GS_OK;
}

extern "C" int gs_ctx_set_stream(gs_ctx *ctx, void *stream) {
    GS_REQUIRE(ctx != nullptr, "gs_ctx_set_stream: ctx is NULL");
    if (const int rc = check_stream_priority(ctx, (hipStream_t)stream)) return rc;
    ctx->stream = (hipStream_t)stream;
    return GS_OK;
}

extern "C" int64_t gs_ctx_info(gs_ctx *ctx, const char *name) {
    if (!ctx || !name) return -1;
    if (strcmp(name, "workspace_bytes") == 0) return (int64_t)ctx->workspace_bytes;
    if (strcmp(name, "allocations") == 0) return ctx->allocations;
    return -1;
}

extern "C" int gs_ctx_use_records(gs_ctx *ctx, int32_t N, const float *records) {
    GS_REQUIRE(ctx, "gs_ctx_use_records: ctx is NULL");
    GS_REQUIRE(((uintptr_t)records & 15) == 0, "gs_ctx_use_records: records must be 16-byte aligned");
    GS_REQUIRE(records == nullptr || N >= 0, "gs_ctx_use_records: N=%d < 0", N);
    ctx->packed_fresh = false;
    ctx->ext_records = reinterpret_cast<const float4 *>(records);
    ctx->ext_records_N = records ? N : -1;
    return GS_OK;
}

extern "C" int gs_ctx_destroy(gs_ctx *ctx) {
    if (!ctx) return GS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    gs_buf *bufs[] = {&ctx->counts, &ctx->offsets, &ctx->block_sums, &ctx->total, &ctx->depth_sort,
                      &ctx->keys_tmp, &ctx->vals_tmp, &ctx->radix_table, &ctx->packed, &ctx->grad_rows,
                      &ctx->tile_scratch, &ctx->loss_partials, &ctx->stats_words, &ctx->det_partials, &ctx->det_sort,
                      &ctx->tile_order_buf, &ctx->bounds, &ctx->tile_counts, &ctx->sorted_aux};
    for (gs_buf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (gs_prof_span &sp : ctx->prof_spans) { (void)hipEventDestroy(sp.start); (void)hipEventDestroy(sp.stop); }
    for (hipEvent_t e : ctx->prof_free) (void)hipEventDestroy(e);
    if (ctx->total_host) (void)hipHostFree(ctx->total_host);
    if (ctx->count_event) (void)hipEventDestroy(ctx->count_event);
    if (ctx->bounds_event) (void)hipEventDestroy(ctx->bounds_event);
    delete ctx;
    return GS_OK;
}
