// K2-K5 binning: replaces match_gaus_to_tiles (rasterization.py:6-25), kernel tile_match
// (setup_gpu.py:23-44) and the per-tile filter + RadixSort (rasterization.py:57,66-71).
//
// The reference materialises a dense u64 [tiles_x, tiles_y, N] table (65 GB at 1M x 1080p) and
// sorts each tile's filtered slice on the host.  Here: tiles-per-gaussian -> exclusive scan ->
// duplicate into (tile_id<<32 | u32(depth), gaussian) pairs in gaussian-major order -> one stable
// LSD radix sort over (32 + ceil(log2 tiles)) key bits -> per-tile [start,end).  A stable sort on
// (tile, depth) of a gaussian-ascending input gives per tile (depth asc, gaussian asc), which is
// exactly the reference's stable low-32-bit sort of (g<<32)+depth (quirk Q12).
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // gaussians per block

struct Grid {
    int tiles_x, tiles_y;
};

// tiles covered by gaussian g: rect clipped to the grid (tile_match only visits grid tiles,
// setup_gpu.py:27-29); u32(depth)==0 entries are filtered out (rasterization.py:67)
__device__ inline uint32_t tile_count(const int4 r, int depth, Grid g, int &x0, int &y0, int &h) {
    x0 = max(r.x, 0);
    y0 = max(r.y, 0);
    const int x1 = min(r.z, g.tiles_x), y1 = min(r.w, g.tiles_y);
    const int w = max(x1 - x0, 0);
    h = max(y1 - y0, 0);
    return depth == 0 ? 0u : (uint32_t)w * (uint32_t)h;
}

__device__ inline uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns the
// exclusive prefix, *block_total gets the sum
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *block_total) {
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const uint32_t s = wave_sums[w];
        if (w < wave) base += s;
        total += s;
    }
    __syncthreads();
    *block_total = total;
    return base + inc - v;
}

// K2a: per-block totals of tiles-per-gaussian
__global__ __launch_bounds__(SCAN_THREADS) void count_kernel(int N, const int4 *__restrict__ rect,
                                                             const int32_t *__restrict__ depth,
                                                             Grid grid, uint32_t *__restrict__ block_sums) {
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int g = base + k;
        if (g < N) {
            int x0, y0, h;
            sum += tile_count(rect[g], depth[g], grid, x0, y0, h);
        }
    }
    uint32_t total;
    block_exclusive_scan(sum, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// K2b: single block turns block_sums into exclusive block offsets (u64 total)
__global__ __launch_bounds__(1024) void scan_blocks_kernel(int nblocks, uint32_t *__restrict__ block_sums,
                                                           uint64_t *__restrict__ total_out) {
    __shared__ uint64_t wave_sums[16];
    __shared__ uint64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0u;
        // wave scan in 64-bit to survive D >= 2^32 checks later
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = (uint64_t)__shfl_up((unsigned long long)inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        uint64_t wbase = 0, chunk_total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) wbase += wave_sums[w];
            chunk_total += wave_sums[w];
        }
        const uint64_t carry = carry_s;
        const uint64_t excl = carry + wbase + inc - v;
        // offsets are stored as u32: D < 2^32 is enforced on the host
        if (i < nblocks) block_sums[i] = (uint32_t)excl;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + chunk_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

// K2c: exclusive offsets per gaussian
__global__ __launch_bounds__(SCAN_THREADS) void offsets_kernel(int N, const int4 *__restrict__ rect,
                                                               const int32_t *__restrict__ depth, Grid grid,
                                                               const uint32_t *__restrict__ block_offsets,
                                                               uint32_t *__restrict__ offsets) {
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int g = base + k;
        c[k] = 0;
        if (g < N) {
            int x0, y0, h;
            c[k] = tile_count(rect[g], depth[g], grid, x0, y0, h);
        }
        sum += c[k];
    }
    uint32_t total;
    uint32_t run = block_offsets[blockIdx.x] + block_exclusive_scan(sum, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int g = base + k;
        if (g < N) offsets[g] = run;
        run += c[k];
    }
}

// K3: duplicate.  One wave per 64 consecutive gaussians; the wave's output slots are a
// contiguous range, filled cooperatively so stores are coalesced and a gaussian covering
// hundreds of tiles does not serialise its wave.
__global__ __launch_bounds__(256) void duplicate_kernel(int N, const int4 *__restrict__ rect,
                                                        const int32_t *__restrict__ depth, Grid grid,
                                                        const uint32_t *__restrict__ offsets,
                                                        uint64_t *__restrict__ keys, int32_t *__restrict__ vals) {
    __shared__ uint32_t s_pref[4][64];   // inclusive prefix of counts within the wave
    __shared__ int4 s_info[4][64];       // x0, y0, h, depth
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = (blockIdx.x * 4 + wave) * 64 + lane;
    uint32_t cnt = 0;
    int x0 = 0, y0 = 0, h = 1, d = 0;
    if (g < N) {
        d = depth[g];
        cnt = tile_count(rect[g], d, grid, x0, y0, h);
    }
    const uint32_t inc = wave_inclusive_scan(cnt);
    s_pref[wave][lane] = inc;
    s_info[wave][lane] = make_int4(x0, y0, h, d);
    const uint32_t wave_total = __shfl(inc, 63, 64);   // 0 when the whole wave is past N
    const int g0 = (blockIdx.x * 4 + wave) * 64;
    const uint32_t out_base = g0 < N ? offsets[g0] : 0u;
    __syncthreads();
    for (uint32_t sidx = lane; sidx < wave_total; sidx += 64) {
        // owner = first lane whose inclusive prefix exceeds sidx (binary search, 6 steps)
        int lo = 0, hi = 63;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_pref[wave][mid] > sidx) hi = mid; else lo = mid + 1;
        }
        const int4 inf = s_info[wave][lo];
        const uint32_t first = lo ? s_pref[wave][lo - 1] : 0u;
        const uint32_t local = sidx - first;
        const int tx = inf.x + (int)(local / (uint32_t)inf.z);
        const int ty = inf.y + (int)(local % (uint32_t)inf.z);
        const uint64_t tile = (uint64_t)(uint32_t)(tx * grid.tiles_y + ty);
        keys[out_base + sidx] = (tile << 32) | (uint64_t)(uint32_t)inf.w;
        vals[out_base + sidx] = g0 + lo;
    }
}

// K5: per-tile [start,end) from the sorted keys
__global__ __launch_bounds__(256) void ranges_kernel(uint32_t D, const uint64_t *__restrict__ keys,
                                                     int32_t *__restrict__ ranges) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= D) return;
    const uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0) {
        ranges[2 * t] = 0;
    } else {
        const uint32_t p = (uint32_t)(keys[i - 1] >> 32);
        if (p != t) {
            ranges[2 * p + 1] = (int32_t)i;
            ranges[2 * t] = (int32_t)i;
        }
    }
    if (i == D - 1) ranges[2 * t + 1] = (int32_t)D;
}

int check_bin_args(gs_ctx *ctx, int32_t N, const int32_t *rect, const int32_t *depths, int32_t res_w,
                   int32_t res_h, int32_t tile_size) {
    GS_REQUIRE(ctx, "gs_bin: ctx is NULL");
    GS_REQUIRE(N >= 0, "gs_bin: N=%d < 0", N);
    GS_REQUIRE(res_w > 0 && res_h > 0 && tile_size > 0, "gs_bin: invalid frame %dx%d tile %d", res_w,
               res_h, tile_size);
    GS_REQUIRE(N == 0 || (rect && depths), "gs_bin: NULL array argument");
    GS_REQUIRE(((uintptr_t)rect & 15) == 0, "gs_bin: tiles_touched must be 16-byte aligned");
    return GS_OK;
}

}  // namespace

extern "C" int gs_bin_count(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                            int32_t res_w, int32_t res_h, int32_t tile_size, int64_t *D_host) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    GS_REQUIRE(D_host, "gs_bin_count: D_host is NULL");
    ctx->bin_N = -1;
    ctx->bin_D = -1;
    *D_host = 0;
    if (N == 0) {
        ctx->bin_N = 0;
        ctx->bin_D = 0;
        return GS_OK;
    }
    GS_HIP(hipSetDevice(ctx->device));
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    const int nblocks = gs_div_up(N, SCAN_TILE);
    if ((rc = gs_buf_reserve(ctx, &ctx->offsets, (size_t)N * 4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->block_sums, (size_t)nblocks * 4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->total, 8))) return rc;
    const int4 *rect = reinterpret_cast<const int4 *>(tiles_touched);
    uint32_t *bs = (uint32_t *)ctx->block_sums.p;
    { gs_prof_scope prof_(ctx, GS_K_COUNT); hipLaunchKernelGGL(count_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, N, rect, depths,
                       grid, bs); }
    { gs_prof_scope prof_(ctx, GS_K_SCAN); hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, nblocks, bs,
                       (uint64_t *)ctx->total.p); }
    { gs_prof_scope prof_(ctx, GS_K_OFFSETS); hipLaunchKernelGGL(offsets_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, N, rect, depths,
                       grid, bs, (uint32_t *)ctx->offsets.p); }
    GS_HIP(hipGetLastError());
    GS_HIP(hipMemcpyAsync(ctx->total_host, ctx->total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t D = *ctx->total_host;
    if (D >= (1ull << 31))
        return gs_set_error(GS_ERR_ARG, "gs_bin_count: %llu duplicates exceed the 2^31-1 limit",
                            (unsigned long long)D);
    ctx->bin_N = N;
    ctx->bin_D = (int64_t)D;
    *D_host = (int64_t)D;
    return GS_OK;
}

extern "C" int gs_bin_fill(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                           int32_t res_w, int32_t res_h, int32_t tile_size, int64_t D, uint64_t *keys,
                           int32_t *vals, int32_t *ranges) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    GS_REQUIRE(ranges, "gs_bin_fill: ranges is NULL");
    if (ctx->bin_N != N || ctx->bin_D != D)
        return gs_set_error(GS_ERR_STATE, "gs_bin_fill: call gs_bin_count first with the same inputs "
                            "(have N=%d D=%lld, got N=%d D=%lld)", ctx->bin_N, (long long)ctx->bin_D, N,
                            (long long)D);
    GS_HIP(hipSetDevice(ctx->device));
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    const int tiles = grid.tiles_x * grid.tiles_y;
    GS_HIP(hipMemsetAsync(ranges, 0, (size_t)tiles * 8, ctx->stream));
    if (D == 0) return GS_OK;
    GS_REQUIRE(keys && vals, "gs_bin_fill: NULL keys/vals with D=%lld", (long long)D);
    if ((rc = gs_buf_reserve(ctx, &ctx->keys_tmp, (size_t)D * 8))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->vals_tmp, (size_t)D * 4))) return rc;
    uint64_t *k_in = (uint64_t *)ctx->keys_tmp.p;
    int32_t *v_in = (int32_t *)ctx->vals_tmp.p;
    { gs_prof_scope prof_(ctx, GS_K_DUPLICATE); hipLaunchKernelGGL(duplicate_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                       reinterpret_cast<const int4 *>(tiles_touched), depths, grid,
                       (const uint32_t *)ctx->offsets.p, k_in, v_in); }
    GS_HIP(hipGetLastError());

    // K4: stable LSD radix sort on the live key bits only
    int tile_bits = 0;
    while ((1 << tile_bits) < tiles) tile_bits++;
    const unsigned end_bit = 32 + (tile_bits ? tile_bits : 1);
    size_t tmp_bytes = 0;
    GS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, keys, v_in, vals, (size_t)D, 0u, end_bit,
                                     ctx->stream));
    if ((rc = gs_buf_reserve(ctx, &ctx->sort_tmp, tmp_bytes))) return rc;
    {
        gs_prof_scope prof_(ctx, GS_K_SORT);
        GS_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp_bytes, k_in, keys, v_in, vals, (size_t)D, 0u,
                                         end_bit, ctx->stream));
    }

    { gs_prof_scope prof_(ctx, GS_K_RANGES); hipLaunchKernelGGL(ranges_kernel, dim3(gs_div_up((int)D, 256)), dim3(256), 0, ctx->stream,
                       (uint32_t)D, keys, ranges); }
    GS_HIP(hipGetLastError());
    return GS_OK;
}
