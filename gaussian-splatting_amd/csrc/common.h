// Internal definitions shared by the translation units of libgsplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/gsplat.h"

#define GS_OK 0
#define GS_ERR_ARG (-1)
#define GS_ERR_HIP (-2)
#define GS_ERR_STATE (-3)

int gs_set_error(int code, const char *fmt, ...);

#define GS_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return gs_set_error(GS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define GS_REQUIRE(cond, ...)                                                              \
    do {                                                                                   \
        if (!(cond)) return gs_set_error(GS_ERR_ARG, __VA_ARGS__);                         \
    } while (0)

// per-kernel HIP-event timing (off by default; see gs_ctx_profile)
enum gs_kernel_id {
    GS_K_PREPROCESS, GS_K_COUNT, GS_K_SCAN, GS_K_OFFSETS, GS_K_DUPLICATE, GS_K_SORT, GS_K_RANGES,
    GS_K_BLEND_FWD, GS_K_BLEND_BWD, GS_K_UNPACK, GS_K_PARAM_BWD, GS_K_PACK, GS_K_DEPTH_SORT, GS_K_LOSS, GS_K_ADAM, GS_K_NUM
};

struct gs_prof_span {
    int kernel;
    hipEvent_t start, stop;
};

// grow-only device buffer owned by the ctx (no hipMalloc in steady state)
struct gs_buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct gs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // binning workspace
    gs_buf counts;       // u32[N]     tiles per gaussian
    gs_buf offsets;      // u32[N]     exclusive scan of tiles-per-gaussian in depth order
    gs_buf block_sums;   // u32[nblocks]
    gs_buf total;        // u64 D, u32 max depth on device
    gs_buf depth_sort;   // 2 x u32[N] keys + 2 x i32[N] values (ping-pong of the depth sort)
    gs_buf keys_tmp;     // 2 x u32[D] tile ids (ping-pong of the tile sort)
    gs_buf vals_tmp;     // i32[D]
    gs_buf radix_table;  // u32[256] digit totals + u32[256][blocks] per-block digit counts
    // blend workspace
    gs_buf packed;       // float4[N][4] 64-byte per-gaussian blend records
    gs_buf grad_rows;    // float[N][16] 64-byte gradient rows for blend backward
    gs_buf tile_scratch; // rasterize_tile shim staging
    gs_buf loss_partials; // f64[1024] per-block loss sums
    gs_buf stats_words;   // u64[8] counters of gs_view_stats
    gs_buf det_partials;  // deterministic backward: float[nsub][D][12] per-(tile, gaussian) partial gradients
    gs_buf det_sort;      // deterministic backward: 2 x u32[D] keys + 2 x i32[D] positions (sort by gaussian)
    gs_buf bounds;        // gs_bin_bounds_batch: per-block partials and the per-view (pairs, max depth) words
    hipEvent_t bounds_event = nullptr;
    uint64_t *total_host = nullptr;   // pinned: 2 words (D, max depth) + 2 per view of a batch
    hipEvent_t count_event = nullptr; // recorded behind the copy of D into total_host (gs_bin_count*_begin)
    int32_t count_pending_N = -1;     // N of a gs_bin_count*_begin whose gs_bin_count_end has not run
    int32_t prepare_pending_views = 0;   // views of a gs_bin_prepare_begin whose gs_bin_prepare_end has not run
    int32_t prepare_N = -1;
    // state carried from gs_bin_count to gs_bin_fill
    bool prof_on = false;
    uint32_t prof_mask = 0xFFFFFFFFu;         // kernel ids that are timed when prof_on (option profile_mask)
    std::vector<gs_prof_span> prof_spans;     // recorded, not yet read
    std::vector<hipEvent_t> prof_free;        // recycled events
    int ppt = 4;         // pixels per thread of the blend kernels (1, 2 or 4)
    bool deterministic_bwd = false;   // option deterministic_backward
    bool tile_order = true;           // option tile_order: blend kernels take their tiles heaviest first
    int tile_order_min = 2048, tile_order_max = 16384;   // ... for frames with this many 16x16 blocks
    gs_buf tile_order_buf;            // i32[ntiles] launch order + i32[ntiles] work per tile
    gs_buf tile_counts;               // u32[tiles] pairs per tile (first pass of the tile sort -> K5)
    const void *tile_counts_clean = nullptr;   // == tile_counts.p while its first tile_counts_tiles counters are zero
    int tile_counts_tiles = 0;
    bool tile_hist = true;            // option tile_histogram: K5 from that histogram instead of the sorted tile ids
    // option sorted_count_from: from this many gaussians on the depth sort runs before the tile count and the count,
    // scan and emit kernels stream depth-ordered arrays instead of gathering (binning_kernels.h, SortedAux)
    int32_t sorted_count_from = 2 << 20;
    bool cull_shrink = true;          // option cull_shrink_rect: culled count / emit walk the ellipse's own bounding box
    bool count_sorted = false;        // the pending / last count of this ctx ran in that form
    int depth_bits_hint = 0;          // key bits the last sorted count on this ctx needed (0: none yet -> sort all 32)
    int sorted_bits = 0;              // key bits the pending sorted count sorted
    bool sorted_redo = false;         // ... which were too few: bin_fill sorts and counts again with all bits
    gs_buf sorted_aux;                // 56 B per gaussian in depth order: clipped rect + id, two record quarters, test mask
                                      // (the unsorted culled count keeps walked rect + test mask here, 24 B by gaussian id)
    int32_t bin_N = -1;
    int64_t bin_D = -1;
    uint32_t bin_max_depth = 0;
    // arrays the packed blend records were built from (gs_blend_forward -> gs_blend_backward reuse)
    const void *packed_key[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t packed_N = -1;
    const float4 *ext_records = nullptr;   // gs_ctx_use_records: caller-owned records replace ctx->packed
    int32_t ext_records_N = -1;
    bool packed_fresh = false;   // gs_bin_fill_culled packed them: the next gs_blend_forward may reuse them
    bool bin_culled = false;     // gs_bin_count_culled produced bin_N / bin_D
    std::vector<hipStream_t> checked_streams;   // streams whose priority has been checked (gs_ctx_set_stream)
    double wait_limit_ms = 30000.0;   // longest host wait for an event before the call reports an error
    size_t workspace_bytes = 0;       // bytes held by the gs_bufs of this ctx
    int64_t allocations = 0;          // hipMallocs made by gs_buf_reserve since the ctx was created
};

int gs_buf_reserve(gs_ctx *ctx, gs_buf *b, size_t bytes, bool exact = false);
// ctx.hip: bounded host wait for an event (error instead of a hang when the stream never gets there)
int gs_wait_event(gs_ctx *ctx, hipEvent_t ev, const char *what);
// blend.hip: looks up this device's copy of the static scratch array (gs_blend_backward_noalloc); device must be current
int gs_scratch_setup(int device);
// binning.hip: stable LSD radix sort of n (u32 key, i32 value) pairs on the low `bits` bits; values start as 0..n-1
// when vals_in is NULL.  keys[2] / vals[2] are ping-pong buffers of n elements each; *k_out / *v_out name the
// buffers that hold the result.
int gs_radix_sort_pairs(gs_ctx *ctx, uint32_t n, const uint32_t *keys_in, const int32_t *vals_in, int bits,
                        uint32_t *keys[2], int32_t *vals[2], const uint32_t **k_out, const int32_t **v_out);

// brackets the launches issued during its lifetime with two events on the ctx stream
struct gs_prof_scope {
    gs_ctx *ctx;
    gs_prof_span span;
    bool live = false;
    gs_prof_scope(gs_ctx *c, int kernel);
    ~gs_prof_scope();
};

static inline int gs_div_up(int a, int b) { return (a + b - 1) / b; }

// ---- per-gaussian blend records (blend.hip packs them; the culled binning reads them too) ----
// one 64-byte record per gaussian, so that staging an entry touches ONE cache line:
//   [0] geo = cx, cy, opacity, tau     tau = log2(1/255 / opacity) - margin
//   [1] con = a', b', c', 0            log2-scaled: power2 = a' dx^2 + b' dx dy + c' dy^2
//   [2] rgb = r, g, b, 0               [3] zero
constexpr float GS_LOG2E = 1.4426950408889634f;
constexpr float GS_ALPHA_MIN = 1.0f / 255.0f;   // rasterize.c:114
constexpr float GS_TAU_MARGIN = 0.02f;          // log2 units: 1.4 % in alpha, far above fp32 error

// the one place that defines a record (pack_kernel and the record-writing form of K1 both call it)
__device__ __forceinline__ void gs_make_record(float cx, float cy, float o, float A, float B, float C, float r,
                                               float g, float b, float4 &geo, float4 &con, float4 &rgb) {
    const float tau = (o > 0.0f) ? (__log2f(GS_ALPHA_MIN / o) - GS_TAU_MARGIN) : 1e30f;
    geo = make_float4(cx, cy, o, tau);
    con = make_float4(-0.5f * GS_LOG2E * A, -GS_LOG2E * B, -0.5f * GS_LOG2E * C, 0.0f);
    rgb = make_float4(r, g, b, 0.0f);
}

// Packs the four arrays into ctx->packed on ctx->stream unless `reuse` and the same arrays were
// packed last; *rec gets the record array.
int gs_pack_records(gs_ctx *ctx, int32_t N, const float *centers, const float *colors, const float *opacities,
                    const float *conics, bool reuse, const float4 **rec);

// max over the pixel box [xlo,xhi]x[ylo,yhi] of power2 >= tau ?  u = cx - x in [ulo,uhi], v likewise.
// Q = alpha u^2 + beta u v + gamma v^2 is convex; outside the origin its box minimum lies on one
// of the four edges.  Conics that are not positive definite, NaNs included, are kept undecided.
// inv2g = 0.5 / gamma and inv2a = 0.5 / alpha depend on the gaussian only: callers that test one gaussian against many
// boxes compute them once (gs_box_halves) and pass them in; same operations, same result.
__device__ __forceinline__ float2 gs_box_halves(const float4 con) { return make_float2(0.5f / -con.z, 0.5f / -con.x); }

__device__ __forceinline__ bool gs_box_may_contribute(const float4 geo, const float4 con, float xlo, float xhi,
                                                      float ylo, float yhi, const float2 halves) {
    const float ulo = geo.x - xhi, uhi = geo.x - xlo, vlo = geo.y - yhi, vhi = geo.y - ylo;
    const float al = -con.x, be = -con.y, ga = -con.z;
    if (geo.w >= 1e29f) return false;        // opacity <= 0 or NaN (tau = 1e30): alpha never reaches 1/255
    // not positive definite (or NaN): the edge minima below assume convexity -> undecided, keep
    if (!(al > 0.0f && ga > 0.0f && be * be < 4.0f * al * ga)) return true;
    float qmin;
    if (ulo <= 0.0f && uhi >= 0.0f && vlo <= 0.0f && vhi >= 0.0f) {
        qmin = 0.0f;
    } else {
        const float inv2g = halves.x, inv2a = halves.y;
        float v = fminf(fmaxf(-be * ulo * inv2g, vlo), vhi);
        float q0 = al * ulo * ulo + (be * ulo + ga * v) * v;
        v = fminf(fmaxf(-be * uhi * inv2g, vlo), vhi);
        float q1 = al * uhi * uhi + (be * uhi + ga * v) * v;
        float u = fminf(fmaxf(-be * vlo * inv2a, ulo), uhi);
        float q2 = ga * vlo * vlo + (be * vlo + al * u) * u;
        u = fminf(fmaxf(-be * vhi * inv2a, ulo), uhi);
        float q3 = ga * vhi * vhi + (be * vhi + al * u) * u;
        qmin = fminf(fminf(q0, q1), fminf(q2, q3));
    }
    // relative slack on qmin covers the rounding of the edge evaluation itself
    return !(-qmin * 0.999f + 1e-4f < geo.w);
}
__device__ __forceinline__ bool gs_box_may_contribute(const float4 geo, const float4 con, float xlo, float xhi,
                                                      float ylo, float yhi) {
    return gs_box_may_contribute(geo, con, xlo, xhi, ylo, yhi, gs_box_halves(con));
}
