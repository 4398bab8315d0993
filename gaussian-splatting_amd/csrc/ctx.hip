// Device context: replaces setup_gpu() (setup_gpu.py:3-6,370).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

int gs_buf_reserve(gs_ctx *ctx, gs_buf *b, size_t bytes) {
    if (bytes <= b->cap) return GS_OK;
    // the old block may still be in use by work queued on the stream
    GS_HIP(hipStreamSynchronize(ctx->stream));
    if (b->p) GS_HIP(hipFree(b->p));
    b->p = nullptr;
    b->cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    GS_HIP(hipMalloc(&b->p, want));
    b->cap = want;
    return GS_OK;
}

extern "C" int gs_ctx_create(int device, void *stream, gs_ctx **out) {
    GS_REQUIRE(out != nullptr, "gs_ctx_create: out is NULL");
    int count = 0;
    GS_HIP(hipGetDeviceCount(&count));
    GS_REQUIRE(device >= 0 && device < count, "gs_ctx_create: device %d out of range (%d devices)",
               device, count);
    GS_HIP(hipSetDevice(device));
    gs_ctx *c = new gs_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    hipError_t e = hipHostMalloc((void **)&c->total_host, sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        delete c;
        return gs_set_error(GS_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
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
    return gs_set_error(GS_ERR_ARG, "gs_ctx_set_option: unknown option '%s'", name);
}

extern "C" int gs_ctx_set_stream(gs_ctx *ctx, void *stream) {
    GS_REQUIRE(ctx != nullptr, "gs_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return GS_OK;
}

extern "C" int gs_ctx_destroy(gs_ctx *ctx) {
    if (!ctx) return GS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    gs_buf *bufs[] = {&ctx->offsets, &ctx->block_sums, &ctx->total, &ctx->keys_tmp, &ctx->vals_tmp,
                      &ctx->sort_tmp, &ctx->packed, &ctx->grad_rows, &ctx->tile_scratch};
    for (gs_buf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (ctx->total_host) (void)hipHostFree(ctx->total_host);
    delete ctx;
    return GS_OK;
}
