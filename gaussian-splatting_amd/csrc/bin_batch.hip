// K2-K5 for several views per launch, without a host wait: the lists of up to 8 views are built by the same kernel
// bodies as gs_bin_count_culled / gs_bin_fill_culled (binning_kernels.h), one grid row per view, and the number of
// pairs of a view never leaves the device -- the sorts run over grids sized by the caller's capacity and every
// workgroup reads its view's count from memory.  The reference builds one view's lists at a time on the host
// (rasterization.py:6-25,57-71); a training step on this GPU is bound by how many short dependent launches its
// views need (about 25 per view through the per-view entry points), not by their bytes.
//
//   gs_bin_bounds_batch   per view: the number of rect pairs (an upper bound of the culled list) and the largest
//                         depth key; ONE host wait for the whole batch
//   gs_bin_culled_batch   count -> depth sort -> scan -> emit -> tile sort -> ranges for every view of the batch,
//                         about 20 launches in all, no host wait; D stays in device memory
#include "radix_sort.h"

namespace {

constexpr int BIN_MAX_VIEWS = GS_BIN_MAX_VIEWS;
constexpr int BOUND_BLOCKS = 256;

template <class T>
struct Ptr8 {
    T *p[BIN_MAX_VIEWS];
};

// ---- bounds: sum of rect sizes and max depth key per view
__global__ __launch_bounds__(256) void bounds_kernel(int N, Ptr8<const int4> rect, Ptr8<const int32_t> depth, Grid grid,
                                                     ulonglong2 *__restrict__ partials) {
    __shared__ uint32_t s_max[4];
    __shared__ unsigned long long s_sum[4];
    const int view = blockIdx.y;
    const int4 *r = rect.p[view];
    const int32_t *dp = depth.p[view];
    unsigned long long c = 0;
    uint32_t md = 0;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < N; g += BOUND_BLOCKS * 256) {
        int x0, y0, h;
        const int d = dp[g];
        const uint32_t cg = tile_count(r[g], d, grid, x0, y0, h);
        c += cg;
        if (cg) md = max(md, (uint32_t)d);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        c += __shfl_xor(c, o, 64);
        md = max(md, (uint32_t)__shfl_xor((int)md, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = md; s_sum[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        unsigned long long sum = 0;
        for (int w = 0; w < 4; w++) { m = max(m, s_max[w]); sum += s_sum[w]; }
        partials[view * BOUND_BLOCKS + blockIdx.x] = make_ulonglong2(sum, m);
    }
}

__global__ __launch_bounds__(BOUND_BLOCKS) void bounds_total_kernel(const ulonglong2 *__restrict__ partials,
                                                                    unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s_sum[BOUND_BLOCKS / 64];
    __shared__ uint32_t s_max[BOUND_BLOCKS / 64];
    const int view = blockIdx.x;
    const ulonglong2 p = partials[view * BOUND_BLOCKS + threadIdx.x];
    unsigned long long sum = p.x;
    uint32_t md = (uint32_t)p.y;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sum += __shfl_xor(sum, o, 64);
        md = max(md, (uint32_t)__shfl_xor((int)md, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = md; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        uint32_t m = 0;
        for (int w = 0; w < BOUND_BLOCKS / 64; w++) { t += s_sum[w]; m = max(m, s_max[w]); }
        out[2 * view] = t;
        out[2 * view + 1] = m;
    }
}

// ---- one grid row per view around the shared bodies
struct PairsArgs {
    const int4 *rect[BIN_MAX_VIEWS];
    const int32_t *depth[BIN_MAX_VIEWS];
    const float4 *rec[BIN_MAX_VIEWS];
    const int32_t *perm[BIN_MAX_VIEWS];
    uint32_t *counts[BIN_MAX_VIEWS];
    const uint32_t *offsets[BIN_MAX_VIEWS];
    uint2 *pairs[BIN_MAX_VIEWS];
};

template <bool EMIT>
__global__ __launch_bounds__(256) void pairs_batch_kernel(PairsArgs a, int N, Grid grid, CullBox cb, uint32_t cap) {
    const int v = blockIdx.y;
    culled_pairs_body<EMIT>(blockIdx.x, N, a.perm[v], a.rect[v], a.depth[v], grid, a.rec[v], cb, a.counts[v],
                            a.offsets[v], cap, a.pairs[v]);
}

// exact lists: tiles per gaussian from the rect alone (the count of gs_bin_count)
__global__ __launch_bounds__(256) void rect_count_batch_kernel(PairsArgs a, int N, Grid grid) {
    const int v = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    int x0, y0, h;
    a.counts[v][g] = tile_count(a.rect[v][g], a.depth[v][g], grid, x0, y0, h);
}

// per view: number of pairs and the largest depth key among the gaussians that own one (partials as bounds_kernel)
__global__ __launch_bounds__(256) void count_sum_batch_kernel(PairsArgs a, int N, ulonglong2 *__restrict__ partials) {
    __shared__ uint32_t s_max[4];
    __shared__ unsigned long long s_sum[4];
    const int view = blockIdx.y;
    const uint32_t *cn = a.counts[view];
    const int32_t *dp = a.depth[view];
    unsigned long long c = 0;
    uint32_t md = 0;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < N; g += BOUND_BLOCKS * 256) {
        const uint32_t cg = cn[g];
        c += cg;
        if (cg) md = max(md, (uint32_t)dp[g]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        c += __shfl_xor(c, o, 64);
        md = max(md, (uint32_t)__shfl_xor((int)md, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = md; s_sum[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        unsigned long long sum = 0;
        for (int w = 0; w < 4; w++) { m = max(m, s_max[w]); sum += s_sum[w]; }
        partials[view * BOUND_BLOCKS + blockIdx.x] = make_ulonglong2(sum, m);
    }
}

struct ScanArgs {
    const int32_t *perm[BIN_MAX_VIEWS];
    const uint32_t *counts[BIN_MAX_VIEWS];
    uint32_t *block_sums[BIN_MAX_VIEWS];
    uint32_t *offsets[BIN_MAX_VIEWS];
    uint32_t *total[BIN_MAX_VIEWS];
};

__global__ __launch_bounds__(SCAN_THREADS) void block_sums_batch_kernel(ScanArgs a, int N) {
    const int v = blockIdx.y;
    sorted_block_sums_body(blockIdx.x, N, a.perm[v], a.counts[v], a.block_sums[v]);
}
__global__ __launch_bounds__(1024) void scan_blocks_batch_kernel(ScanArgs a, int nblocks) {
    const int v = blockIdx.x;
    scan_blocks_body(0, nblocks, a.block_sums[v], a.total[v]);
}
__global__ __launch_bounds__(SCAN_THREADS) void offsets_batch_kernel(ScanArgs a, int N) {
    const int v = blockIdx.y;
    sorted_offsets_body(blockIdx.x, N, a.perm[v], a.counts[v], a.block_sums[v], a.offsets[v]);
}

struct RadixArgs {
    RadixIO io[BIN_MAX_VIEWS];
    uint32_t *table[BIN_MAX_VIEWS];       // [256] digit totals, then [256][nblocks]
    const uint32_t *n_dev[BIN_MAX_VIEWS]; // NULL: every view sorts n elements
    uint32_t n;                           // with n_dev: the capacity (elements beyond it do not exist)
};

__device__ __forceinline__ uint32_t view_n(const RadixArgs &a, int v) {
    return a.n_dev[v] ? min(*a.n_dev[v], a.n) : a.n;
}

// the batch sorts with 4096-element workgroups whatever the capacity: a view's real count is only known on the device
template <bool PAIRS>
__global__ __launch_bounds__(RADIX_THREADS) void hist_batch_kernel(RadixArgs a, int shift, uint32_t mask, uint32_t nblocks) {
    const int v = blockIdx.y;
    radix2_hist_body<RADIX_THREADS, RADIX_ITEMS, PAIRS>(blockIdx.x, view_n(a, v), a.io[v].keys_in, a.io[v].pairs_in, shift, mask,
                                                        nblocks, a.table[v] + RADIX_BINS);
}
__global__ __launch_bounds__(SCAN_THREADS) void rscan_batch_kernel(RadixArgs a, uint32_t nblocks) {
    const int v = blockIdx.y;
    radix_scan_body(blockIdx.x, nblocks, a.table[v] + RADIX_BINS, a.table[v]);
}
template <int NBITS, int MODE>
__global__ __launch_bounds__(RADIX_THREADS) void scatter_batch_kernel(RadixArgs a, int shift, uint32_t mask, uint32_t nblocks) {
    const int v = blockIdx.y;
    radix2_scatter_body<NBITS, RADIX_THREADS, RADIX_ITEMS, MODE>(blockIdx.x, view_n(a, v), a.io[v], shift, mask, nblocks,
                                                                 a.table[v] + RADIX_BINS, a.table[v]);
}

struct RangesArgs {
    const uint32_t *tile_keys[BIN_MAX_VIEWS];
    int32_t *ranges[BIN_MAX_VIEWS];
    const uint32_t *n_dev[BIN_MAX_VIEWS];
    int64_t *D_out[BIN_MAX_VIEWS];
    uint32_t cap;
};
__global__ __launch_bounds__(256) void zero_ranges_batch_kernel(RangesArgs a, int words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < words) a.ranges[blockIdx.y][i] = 0;
}
__global__ __launch_bounds__(256) void ranges_batch_kernel(RangesArgs a) {
    const int v = blockIdx.y;
    const uint32_t total = *a.n_dev[v];
    ranges_body(blockIdx.x, min(total, a.cap), a.tile_keys[v], a.ranges[v]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.D_out[v]) {
        a.D_out[v][0] = (int64_t)min(total, a.cap);     // pairs in the list
        a.D_out[v][1] = (int64_t)total;                 // pairs found: larger than [0] = the capacity was too small
    }
}

int radix_pass_batch(gs_ctx *ctx, int nv, const RadixArgs &a, uint32_t n_upper, int mode, int shift, int nbits) {
    constexpr int TILE = RADIX_THREADS * RADIX_ITEMS;
    const uint32_t nblocks = (n_upper + TILE - 1) / TILE;
    const uint32_t mask = (1u << nbits) - 1u;
    if (mode == RADIX_SOA)
        hipLaunchKernelGGL(hist_batch_kernel<false>, dim3(nblocks, nv), dim3(RADIX_THREADS), 0, ctx->stream, a, shift, mask, nblocks);
    else
        hipLaunchKernelGGL(hist_batch_kernel<true>, dim3(nblocks, nv), dim3(RADIX_THREADS), 0, ctx->stream, a, shift, mask, nblocks);
    hipLaunchKernelGGL(rscan_batch_kernel, dim3(RADIX_BINS, nv), dim3(SCAN_THREADS), 0, ctx->stream, a, nblocks);
#define GS_SCATTER_M(NB, M)                                                                                            \
    hipLaunchKernelGGL((scatter_batch_kernel<NB, M>), dim3(nblocks, nv), dim3(RADIX_THREADS), 0, ctx->stream, a, shift, mask, \
                       nblocks)
#define GS_SCATTER(NB)                                                                     \
    case NB:                                                                               \
        if (mode == RADIX_SOA) GS_SCATTER_M(NB, RADIX_SOA);                                \
        else if (mode == RADIX_PAIRS) GS_SCATTER_M(NB, RADIX_PAIRS);                       \
        else GS_SCATTER_M(NB, RADIX_PAIRS_TO_SOA);                                         \
        break;
    switch (nbits) {
        GS_SCATTER(1) GS_SCATTER(2) GS_SCATTER(3) GS_SCATTER(4) GS_SCATTER(5) GS_SCATTER(6) GS_SCATTER(7) GS_SCATTER(8)
        default: return gs_set_error(GS_ERR_ARG, "radix_pass_batch: %d bits per pass", nbits);
    }
#undef GS_SCATTER
#undef GS_SCATTER_M
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int bits_of_u32(uint32_t v) {
    int b = 0;
    while (v) { b++; v >>= 1; }
    return b;
}

int check_views(gs_ctx *ctx, int32_t N, int32_t n_views, const gs_view_bin *views, int32_t res_w, int32_t res_h,
                int32_t tile_size, const char *who) {
    GS_REQUIRE(ctx, "%s: ctx is NULL", who);
    GS_REQUIRE(N >= 0, "%s: N=%d < 0", who, N);
    GS_REQUIRE(n_views >= 1 && n_views <= BIN_MAX_VIEWS, "%s: n_views=%d not in [1, %d]", who, n_views, BIN_MAX_VIEWS);
    GS_REQUIRE(views, "%s: views is NULL", who);
    GS_REQUIRE(res_w > 0 && res_h > 0 && tile_size > 0, "%s: invalid frame %dx%d tile %d", who, res_w, res_h, tile_size);
    for (int v = 0; v < n_views; v++) {
        GS_REQUIRE(N == 0 || (views[v].tiles_touched && views[v].depths), "%s: view %d has a NULL input array", who, v);
        GS_REQUIRE(((uintptr_t)views[v].tiles_touched & 15) == 0, "%s: tiles_touched must be 16-byte aligned", who);
    }
    return GS_OK;
}

}  // namespace

extern "C" int gs_bin_bounds_batch(gs_ctx *ctx, int32_t N, int32_t n_views, const gs_view_bin *views, int32_t res_w,
                                   int32_t res_h, int32_t tile_size, int64_t *bounds_host) {
    int rc = check_views(ctx, N, n_views, views, res_w, res_h, tile_size, "gs_bin_bounds_batch");
    if (rc) return rc;
    GS_REQUIRE(bounds_host, "gs_bin_bounds_batch: bounds_host is NULL");
    for (int v = 0; v < 2 * n_views; v++) bounds_host[v] = 0;
    if (N == 0) return GS_OK;
    GS_HIP(hipSetDevice(ctx->device));
    const size_t part_bytes = (size_t)BIN_MAX_VIEWS * BOUND_BLOCKS * sizeof(ulonglong2);
    if ((rc = gs_buf_reserve(ctx, &ctx->bounds, part_bytes + 2 * BIN_MAX_VIEWS * 8))) return rc;
    if (!ctx->bounds_event) GS_HIP(hipEventCreateWithFlags(&ctx->bounds_event, hipEventDisableTiming));
    ulonglong2 *partials = (ulonglong2 *)ctx->bounds.p;
    unsigned long long *out = (unsigned long long *)((char *)ctx->bounds.p + part_bytes);
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    Ptr8<const int4> rect{};
    Ptr8<const int32_t> depth{};
    for (int v = 0; v < n_views; v++) {
        rect.p[v] = reinterpret_cast<const int4 *>(views[v].tiles_touched);
        depth.p[v] = views[v].depths;
    }
    {
        gs_prof_scope prof_(ctx, GS_K_COUNT);
        hipLaunchKernelGGL(bounds_kernel, dim3(BOUND_BLOCKS, n_views), dim3(256), 0, ctx->stream, N, rect, depth, grid,
                           partials);
        hipLaunchKernelGGL(bounds_total_kernel, dim3(n_views), dim3(BOUND_BLOCKS), 0, ctx->stream,
                           (const ulonglong2 *)partials, out);
    }
    GS_HIP(hipGetLastError());
    GS_HIP(hipMemcpyAsync(ctx->total_host + 2, out, (size_t)n_views * 16, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipEventRecord(ctx->bounds_event, ctx->stream));
    if ((rc = gs_wait_event(ctx, ctx->bounds_event, "gs_bin_bounds_batch"))) return rc;
    for (int v = 0; v < n_views; v++) {
        const uint64_t pairs = ctx->total_host[2 + 2 * v];
        if (pairs >= (1ull << 31))
            return gs_set_error(GS_ERR_ARG, "gs_bin_bounds_batch: view %d has %llu rect pairs, beyond the 2^31-1 limit", v,
                                (unsigned long long)pairs);
        bounds_host[2 * v] = (int64_t)pairs;
        bounds_host[2 * v + 1] = (int64_t)ctx->total_host[2 + 2 * v + 1];
    }
    return GS_OK;
}

extern "C" int gs_bin_culled_batch(gs_ctx *ctx, int32_t N, int32_t n_views, const gs_view_bin *views, int32_t res_w,
                                   int32_t res_h, int32_t tile_size, float pixel_offset, int32_t depth_bits,
                                   int64_t capacity) {
    int rc = check_views(ctx, N, n_views, views, res_w, res_h, tile_size, "gs_bin_culled_batch");
    if (rc) return rc;
    GS_REQUIRE(depth_bits >= 1 && depth_bits <= 32, "gs_bin_culled_batch: depth_bits=%d not in [1, 32]", depth_bits);
    GS_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "gs_bin_culled_batch: capacity %lld not in [0, 2^31)",
               (long long)capacity);
    for (int v = 0; v < n_views; v++) {
        GS_REQUIRE(views[v].ranges && views[v].D_dev, "gs_bin_culled_batch: view %d has a NULL ranges / D_dev", v);
        GS_REQUIRE(N == 0 || views[v].records, "gs_bin_culled_batch: view %d has no records", v);
        GS_REQUIRE(capacity == 0 || views[v].vals, "gs_bin_culled_batch: view %d has a NULL vals", v);
        GS_REQUIRE(((uintptr_t)views[v].records & 15) == 0, "gs_bin_culled_batch: records must be 16-byte aligned");
    }
    // the batch overwrites the per-view workspace (counts, offsets, depth order, totals): a pending
    // gs_bin_count*_begin would be answered from another view's numbers, and a later gs_bin_fill* must not pass
    // its state check against what an earlier gs_bin_count left behind
    if (ctx->count_pending_N >= 0 || ctx->prepare_pending_views > 0)
        return gs_set_error(GS_ERR_STATE, "gs_bin_culled_batch: a gs_bin_count*_begin is pending on this ctx "
                            "(call gs_bin_count_end first, or use another ctx)");
    GS_HIP(hipSetDevice(ctx->device));
    ctx->packed_fresh = false;
    ctx->bin_N = -1;
    ctx->bin_D = -1;
    const int nv = n_views;
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    const int tiles = grid.tiles_x * grid.tiles_y;
    const uint32_t cap = (uint32_t)capacity;

    // ---- workspace: one slice per view of every per-view array of the single-view path
    const size_t n4 = ((size_t)(N > 0 ? N : 1) * 4 + 255) & ~(size_t)255, d4 = ((size_t)(cap > 0 ? cap : 1) * 4 + 255) & ~(size_t)255;
    const int scan_blocks = gs_div_up(N > 0 ? N : 1, SCAN_TILE);
    const size_t bs4 = ((size_t)scan_blocks * 4 + 255) & ~(size_t)255;
    const uint32_t nb_max = (uint32_t)gs_div_up((int)((uint32_t)N > cap ? (uint32_t)N : cap), RADIX_THREADS * RADIX_ITEMS) + 1;
    const size_t tab4 = (((size_t)RADIX_BINS * nb_max + RADIX_BINS) * 4 + 255) & ~(size_t)255;
    if ((rc = gs_buf_reserve(ctx, &ctx->counts, nv * n4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->offsets, nv * n4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->block_sums, nv * bs4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->depth_sort, nv * 4 * n4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->keys_tmp, nv * 4 * d4))) return rc;     // pairs ping-pong: 2 x cap x 8 bytes per view
    if ((rc = gs_buf_reserve(ctx, &ctx->vals_tmp, nv * d4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->radix_table, nv * tab4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->total, 16 + (size_t)BIN_MAX_VIEWS * 4))) return rc;
    uint32_t *d_words = (uint32_t *)((char *)ctx->total.p + 16);
    auto at = [](gs_buf &b, size_t off) { return (char *)b.p + off; };

    RangesArgs ra{};
    ra.cap = cap;
    for (int v = 0; v < nv; v++) {
        ra.ranges[v] = views[v].ranges;
        ra.n_dev[v] = d_words + v;
        ra.D_out[v] = views[v].D_dev;
    }
    GS_HIP(hipMemsetAsync(d_words, 0, BIN_MAX_VIEWS * 4, ctx->stream));
    hipLaunchKernelGGL(zero_ranges_batch_kernel, dim3(gs_div_up(2 * tiles, 256), nv), dim3(256), 0, ctx->stream, ra,
                       2 * tiles);
    if (N == 0 || cap == 0) {
        // no gaussian (or no room for a pair): empty lists; D_dev = (0, 0)
        for (int v = 0; v < nv; v++) GS_HIP(hipMemsetAsync(views[v].D_dev, 0, 16, ctx->stream));
        GS_HIP(hipGetLastError());
        return GS_OK;
    }

    const CullBox cb{(float)tile_size, pixel_offset, ctx->cull_shrink ? 1 : 0};
    PairsArgs pa{};
    for (int v = 0; v < nv; v++) {
        pa.rect[v] = reinterpret_cast<const int4 *>(views[v].tiles_touched);
        pa.depth[v] = views[v].depths;
        pa.rec[v] = reinterpret_cast<const float4 *>(views[v].records);
        pa.counts[v] = (uint32_t *)at(ctx->counts, v * n4);
        pa.offsets[v] = (const uint32_t *)at(ctx->offsets, v * n4);
    }
    {   // K2: pairs that pass the box test, per gaussian
        gs_prof_scope prof_(ctx, GS_K_COUNT);
        hipLaunchKernelGGL(pairs_batch_kernel<false>, dim3(gs_div_up(N, 256), nv), dim3(256), 0, ctx->stream, pa, N, grid, cb,
                           0u);
    }
    GS_HIP(hipGetLastError());

    // K4a: gaussians into (u32(depth), id) order
    const int depth_passes = (depth_bits + 7) / 8;
    const int32_t *perm[BIN_MAX_VIEWS] = {};
    {
        gs_prof_scope prof_(ctx, GS_K_DEPTH_SORT);
        for (int p = 0; p < depth_passes; p++) {
            const bool last = p == depth_passes - 1;
            const int nbits = last ? depth_bits - 8 * p : 8;
            RadixArgs a{};
            a.n = (uint32_t)N;
            for (int v = 0; v < nv; v++) {
                char *ds = at(ctx->depth_sort, v * 4 * n4);
                uint32_t *dk[2] = {(uint32_t *)ds, (uint32_t *)(ds + n4)};
                int32_t *dv[2] = {(int32_t *)(ds + 2 * n4), (int32_t *)(ds + 3 * n4)};
                a.io[v].keys_in = p == 0 ? reinterpret_cast<const uint32_t *>(views[v].depths) : dk[(p - 1) & 1];
                a.io[v].vals_in = p == 0 ? nullptr : dv[(p - 1) & 1];
                a.io[v].keys_out = last ? nullptr : dk[p & 1];
                a.io[v].vals_out = dv[p & 1];
                a.table[v] = (uint32_t *)at(ctx->radix_table, v * tab4);
                perm[v] = dv[p & 1];
            }
            if ((rc = radix_pass_batch(ctx, nv, a, (uint32_t)N, RADIX_SOA, 8 * p, nbits))) return rc;
        }
    }

    // K2': exclusive offsets in depth order; the grand total of a view is its D
    ScanArgs sa{};
    for (int v = 0; v < nv; v++) {
        sa.perm[v] = perm[v];
        sa.counts[v] = pa.counts[v];
        sa.block_sums[v] = (uint32_t *)at(ctx->block_sums, v * bs4);
        sa.offsets[v] = (uint32_t *)at(ctx->offsets, v * n4);
        sa.total[v] = d_words + v;
    }
    {
        gs_prof_scope prof_(ctx, GS_K_SCAN);
        hipLaunchKernelGGL(block_sums_batch_kernel, dim3(scan_blocks, nv), dim3(SCAN_THREADS), 0, ctx->stream, sa, N);
        hipLaunchKernelGGL(scan_blocks_batch_kernel, dim3(nv), dim3(1024), 0, ctx->stream, sa, scan_blocks);
        hipLaunchKernelGGL(offsets_batch_kernel, dim3(scan_blocks, nv), dim3(SCAN_THREADS), 0, ctx->stream, sa, N);
    }
    GS_HIP(hipGetLastError());

    // K3 + K4b: pairs in depth order (never past the capacity), then a stable sort by tile id
    int tile_bits = bits_of_u32((uint32_t)(tiles - 1));
    if (tile_bits == 0) tile_bits = 1;
    const int tile_passes = (tile_bits + 7) / 8;
    const int bits_per_pass = (tile_bits + tile_passes - 1) / tile_passes;
    uint2 *pp[BIN_MAX_VIEWS][2];
    uint32_t *sorted_tiles[BIN_MAX_VIEWS];
    for (int v = 0; v < nv; v++) {
        pp[v][0] = (uint2 *)at(ctx->keys_tmp, v * 4 * d4);
        pp[v][1] = (uint2 *)at(ctx->keys_tmp, v * 4 * d4 + 2 * d4);
        sorted_tiles[v] = (uint32_t *)at(ctx->vals_tmp, v * d4);
        pa.perm[v] = perm[v];
        pa.pairs[v] = pp[v][0];
    }
    {
        gs_prof_scope prof_(ctx, GS_K_DUPLICATE);
        hipLaunchKernelGGL(pairs_batch_kernel<true>, dim3(gs_div_up(N, 256), nv), dim3(256), 0, ctx->stream, pa, N, grid, cb,
                           cap);
    }
    GS_HIP(hipGetLastError());
    {
        gs_prof_scope prof_(ctx, GS_K_SORT);
        for (int p = 0; p < tile_passes; p++) {
            const int shift = p * bits_per_pass;
            const int nbits = (tile_bits - shift < bits_per_pass) ? tile_bits - shift : bits_per_pass;
            const bool last = p == tile_passes - 1;
            RadixArgs a{};
            a.n = cap;
            for (int v = 0; v < nv; v++) {
                a.io[v].pairs_in = pp[v][p & 1];
                a.io[v].pairs_out = pp[v][(p & 1) ^ 1];
                a.io[v].keys_out = sorted_tiles[v];
                a.io[v].vals_out = views[v].vals;
                a.table[v] = (uint32_t *)at(ctx->radix_table, v * tab4);
                a.n_dev[v] = d_words + v;
            }
            if ((rc = radix_pass_batch(ctx, nv, a, cap, last ? RADIX_PAIRS_TO_SOA : RADIX_PAIRS, shift, nbits))) return rc;
        }
    }
    {
        gs_prof_scope prof_(ctx, GS_K_RANGES);
        for (int v = 0; v < nv; v++) ra.tile_keys[v] = sorted_tiles[v];
        hipLaunchKernelGGL(ranges_batch_kernel, dim3(gs_div_up((int)cap, 256), nv), dim3(256), 0, ctx->stream, ra);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The front half of the lists of a BATCH of views -- count, depth order, scanned offsets -- in launches that cover
// every view, so that a step needs them once, not once per view (12 short dependent launches and one host wait per
// view otherwise, which the lanes of a training step interleave with their blend kernels).  The back half (emit,
// tile sort, ranges: what scales with D) stays per view: gs_bin_fill_prepared on the view's own ctx / stream.
//   gs_bin_prepare_begin  counts (box test when records are given, the rect alone otherwise), per-view totals,
//                         copy to pinned memory, event; returns at once
//   gs_bin_prepare_end    waits for that copy (bounded), then depth sort and scan for every view into the caller's
//                         perm / offsets; D_host[v] = pairs of view v
namespace {

int prepare_workspace(gs_ctx *ctx, int32_t N, int nv, size_t &n4, size_t &bs4, size_t &tab4, int &scan_blocks) {
    int rc;
    n4 = ((size_t)(N > 0 ? N : 1) * 4 + 255) & ~(size_t)255;
    scan_blocks = gs_div_up(N > 0 ? N : 1, SCAN_TILE);
    bs4 = ((size_t)scan_blocks * 4 + 255) & ~(size_t)255;
    const uint32_t nb = (uint32_t)gs_div_up(N > 0 ? N : 1, RADIX_THREADS * RADIX_ITEMS) + 1;
    tab4 = (((size_t)RADIX_BINS * nb + RADIX_BINS) * 4 + 255) & ~(size_t)255;
    if ((rc = gs_buf_reserve(ctx, &ctx->counts, nv * n4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->block_sums, nv * bs4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->depth_sort, nv * 4 * n4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->radix_table, nv * tab4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->total, 16 + (size_t)BIN_MAX_VIEWS * 4))) return rc;
    const size_t part_bytes = (size_t)BIN_MAX_VIEWS * BOUND_BLOCKS * sizeof(ulonglong2);
    if ((rc = gs_buf_reserve(ctx, &ctx->bounds, part_bytes + 2 * BIN_MAX_VIEWS * 8))) return rc;
    return GS_OK;
}

int check_prepare(gs_ctx *ctx, int32_t N, int32_t n_views, const gs_view_prepare *views, int32_t res_w, int32_t res_h,
                  int32_t tile_size, const char *who) {
    GS_REQUIRE(ctx, "%s: ctx is NULL", who);
    GS_REQUIRE(N >= 0, "%s: N=%d < 0", who, N);
    GS_REQUIRE(n_views >= 1 && n_views <= BIN_MAX_VIEWS, "%s: n_views=%d not in [1, %d]", who, n_views, BIN_MAX_VIEWS);
    GS_REQUIRE(views, "%s: views is NULL", who);
    GS_REQUIRE(res_w > 0 && res_h > 0 && tile_size > 0, "%s: invalid frame %dx%d tile %d", who, res_w, res_h, tile_size);
    for (int v = 0; v < n_views; v++) {
        GS_REQUIRE(N == 0 || (views[v].tiles_touched && views[v].depths && views[v].perm && views[v].offsets),
                   "%s: view %d has a NULL array", who, v);
        GS_REQUIRE(((uintptr_t)views[v].tiles_touched & 15) == 0 && ((uintptr_t)views[v].records & 15) == 0,
                   "%s: tiles_touched and records must be 16-byte aligned", who);
        GS_REQUIRE((views[v].records != nullptr) == (views[0].records != nullptr),
                   "%s: either every view of a batch has records (culled lists) or none has", who);
    }
    return GS_OK;
}

}  // namespace

extern "C" int gs_bin_prepare_begin(gs_ctx *ctx, int32_t N, int32_t n_views, const gs_view_prepare *views, int32_t res_w,
                                    int32_t res_h, int32_t tile_size, float pixel_offset) {
    int rc = check_prepare(ctx, N, n_views, views, res_w, res_h, tile_size, "gs_bin_prepare_begin");
    if (rc) return rc;
    if (ctx->count_pending_N >= 0 || ctx->prepare_pending_views > 0)
        return gs_set_error(GS_ERR_STATE, "gs_bin_prepare_begin: a count is already pending on this ctx");
    GS_HIP(hipSetDevice(ctx->device));
    ctx->packed_fresh = false;
    ctx->bin_N = -1;
    ctx->bin_D = -1;
    ctx->prepare_pending_views = n_views;
    ctx->prepare_N = N;
    if (N == 0) return GS_OK;
    size_t n4, bs4, tab4;
    int scan_blocks;
    if ((rc = prepare_workspace(ctx, N, n_views, n4, bs4, tab4, scan_blocks))) { ctx->prepare_pending_views = 0; return rc; }
    if (!ctx->bounds_event) GS_HIP(hipEventCreateWithFlags(&ctx->bounds_event, hipEventDisableTiming));
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    const CullBox cb{(float)tile_size, pixel_offset, ctx->cull_shrink ? 1 : 0};
    PairsArgs pa{};
    for (int v = 0; v < n_views; v++) {
        pa.rect[v] = reinterpret_cast<const int4 *>(views[v].tiles_touched);
        pa.depth[v] = views[v].depths;
        pa.rec[v] = reinterpret_cast<const float4 *>(views[v].records);
        pa.counts[v] = (uint32_t *)((char *)ctx->counts.p + v * n4);
    }
    const size_t part_bytes = (size_t)BIN_MAX_VIEWS * BOUND_BLOCKS * sizeof(ulonglong2);
    ulonglong2 *partials = (ulonglong2 *)ctx->bounds.p;
    unsigned long long *out = (unsigned long long *)((char *)ctx->bounds.p + part_bytes);
    {
        gs_prof_scope prof_(ctx, GS_K_COUNT);
        if (views[0].records)
            hipLaunchKernelGGL(pairs_batch_kernel<false>, dim3(gs_div_up(N, 256), n_views), dim3(256), 0, ctx->stream, pa, N,
                               grid, cb, 0u);
        else
            hipLaunchKernelGGL(rect_count_batch_kernel, dim3(gs_div_up(N, 256), n_views), dim3(256), 0, ctx->stream, pa, N,
                               grid);
        hipLaunchKernelGGL(count_sum_batch_kernel, dim3(BOUND_BLOCKS, n_views), dim3(256), 0, ctx->stream, pa, N, partials);
        hipLaunchKernelGGL(bounds_total_kernel, dim3(n_views), dim3(BOUND_BLOCKS), 0, ctx->stream,
                           (const ulonglong2 *)partials, out);
    }
    GS_HIP(hipGetLastError());
    GS_HIP(hipMemcpyAsync(ctx->total_host + 2, out, (size_t)n_views * 16, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipEventRecord(ctx->bounds_event, ctx->stream));
    return GS_OK;
}

extern "C" int gs_bin_prepare_end(gs_ctx *ctx, int32_t N, int32_t n_views, const gs_view_prepare *views, int32_t res_w,
                                  int32_t res_h, int32_t tile_size, int64_t *D_host) {
    int rc = check_prepare(ctx, N, n_views, views, res_w, res_h, tile_size, "gs_bin_prepare_end");
    if (rc) return rc;
    GS_REQUIRE(D_host, "gs_bin_prepare_end: D_host is NULL");
    if (ctx->prepare_pending_views != n_views || ctx->prepare_N != N)
        return gs_set_error(GS_ERR_STATE, "gs_bin_prepare_end: no gs_bin_prepare_begin of %d views / %d gaussians is pending "
                            "on this ctx", n_views, N);
    ctx->prepare_pending_views = 0;
    for (int v = 0; v < n_views; v++) D_host[v] = 0;
    if (N == 0) return GS_OK;
    GS_HIP(hipSetDevice(ctx->device));
    if ((rc = gs_wait_event(ctx, ctx->bounds_event, "gs_bin_prepare_end"))) return rc;
    uint32_t max_depth = 0;
    for (int v = 0; v < n_views; v++) {
        const uint64_t D = ctx->total_host[2 + 2 * v];
        if (D >= (1ull << 31))
            return gs_set_error(GS_ERR_ARG, "gs_bin_prepare_end: view %d has %llu pairs, beyond the 2^31-1 limit", v,
                                (unsigned long long)D);
        D_host[v] = (int64_t)D;
        const uint32_t md = (uint32_t)ctx->total_host[2 + 2 * v + 1];
        if (md > max_depth) max_depth = md;
    }
    size_t n4, bs4, tab4;
    int scan_blocks;
    if ((rc = prepare_workspace(ctx, N, n_views, n4, bs4, tab4, scan_blocks))) return rc;   // sized by _begin: no growth
    auto at = [](gs_buf &b, size_t off) { return (char *)b.p + off; };
    const int nv = n_views;
    // K4a: gaussians into (u32(depth), id) order; only the bytes the batch's largest key occupies are sorted
    int depth_bits = bits_of_u32(max_depth);
    if (depth_bits < 1) depth_bits = 1;
    const int depth_passes = (depth_bits + 7) / 8;
    {
        gs_prof_scope prof_(ctx, GS_K_DEPTH_SORT);
        for (int p = 0; p < depth_passes; p++) {
            const bool last = p == depth_passes - 1;
            const int nbits = last ? depth_bits - 8 * p : 8;
            RadixArgs a{};
            a.n = (uint32_t)N;
            for (int v = 0; v < nv; v++) {
                char *ds = at(ctx->depth_sort, v * 4 * n4);
                uint32_t *dk[2] = {(uint32_t *)ds, (uint32_t *)(ds + n4)};
                int32_t *dv[2] = {(int32_t *)(ds + 2 * n4), (int32_t *)(ds + 3 * n4)};
                a.io[v].keys_in = p == 0 ? reinterpret_cast<const uint32_t *>(views[v].depths) : dk[(p - 1) & 1];
                a.io[v].vals_in = p == 0 ? nullptr : dv[(p - 1) & 1];
                a.io[v].keys_out = last ? nullptr : dk[p & 1];
                a.io[v].vals_out = last ? views[v].perm : dv[p & 1];     // the order lands in the caller's array
                a.table[v] = (uint32_t *)at(ctx->radix_table, v * tab4);
            }
            if ((rc = radix_pass_batch(ctx, nv, a, (uint32_t)N, RADIX_SOA, 8 * p, nbits))) return rc;
        }
    }
    // K2': exclusive offsets in depth order, into the caller's array
    uint32_t *d_words = (uint32_t *)((char *)ctx->total.p + 16);
    ScanArgs sa{};
    for (int v = 0; v < nv; v++) {
        sa.perm[v] = views[v].perm;
        sa.counts[v] = (const uint32_t *)at(ctx->counts, v * n4);
        sa.block_sums[v] = (uint32_t *)at(ctx->block_sums, v * bs4);
        sa.offsets[v] = views[v].offsets;
        sa.total[v] = d_words + v;
    }
    {
        gs_prof_scope prof_(ctx, GS_K_SCAN);
        hipLaunchKernelGGL(block_sums_batch_kernel, dim3(scan_blocks, nv), dim3(SCAN_THREADS), 0, ctx->stream, sa, N);
        hipLaunchKernelGGL(scan_blocks_batch_kernel, dim3(nv), dim3(1024), 0, ctx->stream, sa, scan_blocks);
        hipLaunchKernelGGL(offsets_batch_kernel, dim3(scan_blocks, nv), dim3(SCAN_THREADS), 0, ctx->stream, sa, N);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}
