// K2-K5 binning: replaces match_gaus_to_tiles (rasterization.py:6-25), kernel tile_match
// (setup_gpu.py:23-44) and the per-tile filter + RadixSort (rasterization.py:57,66-71).
//
// The reference materialises a dense u64 [tiles_x, tiles_y, N] table (65 GB at 1M x 1080p) and
// sorts each tile's filtered slice on the host: per tile the order is (u32(depth) asc, gaussian
// asc) -- a stable sort on the low 32 bits of (g<<32)+depth over a gaussian-ascending list (quirk
// Q12).  Here the same order is produced in O(N + D) HBM traffic by splitting the 64-bit
// (tile, depth) key into its two fields and sorting each where it is cheapest:
//
//   K2  count        tiles-per-gaussian, D = sum, max u32(depth)          (N elements)
//   K4a depth sort   stable LSD radix sort of the N gaussians by u32(depth), only the bytes that
//                    max(depth) occupies (2 passes for depth < 65536)     (N elements, 8 B each)
//   K2' scan         exclusive offsets in depth order
//   K3  duplicate    (tile_id, gaussian) pairs emitted in depth order     (D elements, 8 B each)
//   K4b tile sort    stable LSD radix sort by tile_id only: ceil(log2 tiles) bits, 2 passes at
//                    1080p instead of the 6 passes over 12-byte pairs a 45-bit key needs
//   K5  ranges       per-tile [start,end); optional 64-bit keys (tile<<32 | u32(depth))
//
// A stable sort by tile of a list that is already in (depth, gaussian) order is in
// (tile, depth, gaussian) order: exactly the reference's per-tile gaus_stack.
//
// The radix sort is hand-written for wave64 (radix_sort.h): per pass one histogram kernel (LDS atomics), one scan
// kernel (one workgroup per digit turns the [digit][block] table into global offsets) and one scatter kernel that
// ranks 4096 or 8192 elements per workgroup with ballot matching (stable inside a wave, per-wave counters in LDS
// across waves), reorders them through LDS and writes each digit run contiguously.  The (tile id, gaussian) pairs
// travel as ONE 8-byte element, so a digit run is one piece of HBM, not two; the last pass splits them into the
// caller's gaussian ids and the sorted tile ids K5 reads.
#include "radix_sort.h"
#include <cstring>

namespace {

// K2: tiles-per-gaussian; per-block partial sums and the largest u32(depth) among the gaussians
// that emit anything (it decides how many depth-sort passes are needed).  COUNT_BLOCKS blocks
// grid-stride over N; count_total_kernel folds the partials (no same-address atomics).
constexpr int COUNT_BLOCKS = 1024;
constexpr int TILE_HIST_MAX = 32768;     // tile ids whose counters fit one workgroup's LDS (128 KiB of the CU's 160)

__global__ __launch_bounds__(SCAN_THREADS) void count_kernel(int N, const int4 *__restrict__ rect,
                                                             const int32_t *__restrict__ depth, Grid grid,
                                                             uint32_t *__restrict__ counts,
                                                             ulonglong2 *__restrict__ partials, int4 *__restrict__ info) {
    __shared__ uint32_t s_max[SCAN_THREADS / 64];
    __shared__ unsigned long long s_sum[SCAN_THREADS / 64];
    unsigned long long c = 0;
    uint32_t md = 0;
    for (int g = blockIdx.x * SCAN_THREADS + threadIdx.x; g < N; g += COUNT_BLOCKS * SCAN_THREADS) {
        int x0, y0, h;
        const int d = depth[g];
        const uint32_t cg = tile_count(rect[g], d, grid, x0, y0, h);
        counts[g] = cg;
        // the clipped rect (empty when the depth key filters the gaussian out): K3 gathers this one item per gaussian
        // instead of rect and depth
        if (info) info[g] = make_int4(x0 | (int)((cg ? cg / (uint32_t)h : 0u) << 16), y0, h, g);
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
        for (int w = 0; w < SCAN_THREADS / 64; w++) { m = max(m, s_max[w]); sum += s_sum[w]; }
        partials[blockIdx.x] = make_ulonglong2(sum, m);
    }
}

// K2 in depth order (large scenes, SortedAux in binning_kernels.h): counts[i] and the clipped rect of gaussian perm[i]
__global__ __launch_bounds__(SCAN_THREADS) void count_sorted_kernel(int N, const int32_t *__restrict__ perm,
                                                                    const int4 *__restrict__ rect,
                                                                    const int32_t *__restrict__ depth, Grid grid,
                                                                    uint32_t *__restrict__ counts, int4 *__restrict__ info,
                                                                    ulonglong2 *__restrict__ partials,
                                                                    const uint32_t *__restrict__ depth_sorted) {
    __shared__ uint32_t s_max[SCAN_THREADS / 64];
    __shared__ unsigned long long s_sum[SCAN_THREADS / 64];
    unsigned long long c = 0;
    uint32_t md = 0;
    for (int i = blockIdx.x * SCAN_THREADS + threadIdx.x; i < N; i += COUNT_BLOCKS * SCAN_THREADS) {
        const int g = perm[i];
        int x0, y0, h;
        const int d = depth_sorted ? (int)depth_sorted[i] : depth[g];
        const uint32_t cg = tile_count(rect[g], d, grid, x0, y0, h);
        counts[i] = cg;
        info[i] = make_int4(x0 | (int)((cg ? cg / (uint32_t)h : 0u) << 16), y0, h, g);
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
        for (int w = 0; w < SCAN_THREADS / 64; w++) { m = max(m, s_max[w]); sum += s_sum[w]; }
        partials[blockIdx.x] = make_ulonglong2(sum, m);
    }
}

__global__ __launch_bounds__(COUNT_BLOCKS) void count_total_kernel(const ulonglong2 *__restrict__ partials,
                                                                   unsigned long long *__restrict__ total) {
    __shared__ unsigned long long s_sum[COUNT_BLOCKS / 64];
    __shared__ uint32_t s_max[COUNT_BLOCKS / 64];
    const ulonglong2 p = partials[threadIdx.x];
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
        for (int w = 0; w < COUNT_BLOCKS / 64; w++) { t += s_sum[w]; m = max(m, s_max[w]); }
        total[0] = t;
        total[1] = m;
    }
}

// K3: duplicate.  One wave per 64 consecutive positions of the depth order; the wave's output
// slots are a contiguous range, filled cooperatively (64 slots per step, slot_owner) so stores are
// coalesced and a gaussian covering hundreds of tiles does not serialise its wave.
__global__ __launch_bounds__(256) void duplicate_kernel(int N, const int32_t *__restrict__ perm,
                                                        const int4 *__restrict__ rect,
                                                        const int32_t *__restrict__ depth, Grid grid,
                                                        const uint32_t *__restrict__ offsets, uint32_t D,
                                                        uint2 *__restrict__ pairs, const int4 *__restrict__ info,
                                                        int info_by_id) {
    __shared__ uint32_t s_first[4][64];  // exclusive prefix of counts within the wave
    __shared__ int4 s_info[4][64];       // x0, y0, h, gaussian
    __shared__ uint32_t s_mark[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wave) * 64;
    const int i = i0 + lane;
    uint32_t cnt = 0;
    int x0 = 0, y0 = 0, h = 1, g = 0;
    if (i < N && info) {             // the clipped rects left by the count: in depth order (large scenes) or by id
        const int4 inf = info[info_by_id ? perm[i] : i];
        x0 = inf.x & 0xFFFF; y0 = inf.y; h = inf.z; g = inf.w;
        cnt = (uint32_t)(inf.x >> 16) * (uint32_t)h;
    } else if (i < N) {
        g = perm[i];
        // perm may be the caller's array (gs_bin_fill_prepared): an id that is not a gaussian owns no pair
        if ((uint32_t)g < (uint32_t)N) cnt = tile_count(rect[g], depth[g], grid, x0, y0, h);
        else g = 0;
    }
    const uint32_t inc = wave_inclusive_scan(cnt);
    const uint32_t first = inc - cnt;
    s_first[wave][lane] = first;
    s_info[wave][lane] = make_int4(x0, y0, h, g);
    const uint32_t wave_total = __shfl(inc, 63, 64);   // 0 when the whole wave is past N
    const uint32_t out_base = i0 < N ? offsets[i0] : 0u;
    uint32_t carry = 0;
    for (uint32_t sbase = 0; sbase < wave_total; sbase += 64) {      // wave-uniform trip count
        const int lo = (int)slot_owner(s_mark[wave], lane, cnt, first, sbase, carry);
        const uint32_t sidx = sbase + lane;
        if (sidx < wave_total && out_base + sidx < D) {     // the second test only fails for offsets of another view
            const int4 inf = s_info[wave][lo];
            const uint32_t local = sidx - s_first[wave][lo];
            const int tx = inf.x + (int)(local / (uint32_t)inf.z);
            const int ty = inf.y + (int)(local % (uint32_t)inf.z);
            pairs[out_base + sidx] = make_uint2((uint32_t)(tx * grid.tiles_y + ty), (uint32_t)inf.w);
        }
    }
}

// D and max depth from the culled counts (same partial layout as count_kernel)
__global__ __launch_bounds__(SCAN_THREADS) void count_sum_kernel(int N, const uint32_t *__restrict__ counts,
                                                                 const int32_t *__restrict__ depth,
                                                                 ulonglong2 *__restrict__ partials) {
    __shared__ uint32_t s_max[SCAN_THREADS / 64];
    __shared__ unsigned long long s_sum[SCAN_THREADS / 64];
    unsigned long long c = 0;
    uint32_t md = 0;
    for (int g = blockIdx.x * SCAN_THREADS + threadIdx.x; g < N; g += COUNT_BLOCKS * SCAN_THREADS) {
        const uint32_t cg = counts[g];
        c += cg;
        if (cg) md = max(md, (uint32_t)depth[g]);
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
        for (int w = 0; w < SCAN_THREADS / 64; w++) { m = max(m, s_max[w]); sum += s_sum[w]; }
        partials[blockIdx.x] = make_ulonglong2(sum, m);
    }
}

// ------------------------------------------------------------- per-view launches of the shared kernel bodies
__global__ __launch_bounds__(SCAN_THREADS) void sorted_block_sums_kernel(int N, const int32_t *__restrict__ perm,
                                                                         const uint32_t *__restrict__ counts,
                                                                         uint32_t *__restrict__ block_sums,
                                                                         uint32_t *__restrict__ counts_sorted) {
    sorted_block_sums_body(blockIdx.x, N, perm, counts, block_sums, counts_sorted);
}
__global__ __launch_bounds__(1024) void scan_blocks_kernel(int nblocks, uint32_t *__restrict__ block_sums) {
    scan_blocks_body(0, nblocks, block_sums, nullptr);
}
__global__ __launch_bounds__(SCAN_THREADS) void sorted_offsets_kernel(int N, const int32_t *__restrict__ perm,
                                                                      const uint32_t *counts,
                                                                      const uint32_t *__restrict__ block_offsets,
                                                                      uint32_t *offsets, int raw_sums) {
    sorted_offsets_body(blockIdx.x, N, perm, counts, block_offsets, offsets, raw_sums != 0);
}
template <int MODE>
__global__ __launch_bounds__(256) void culled_pairs_kernel(int N, const int32_t *__restrict__ perm,
                                                           const int4 *__restrict__ rect,
                                                           const int32_t *__restrict__ depth, Grid grid,
                                                           const float4 *__restrict__ rec, CullBox cb,
                                                           uint32_t *__restrict__ counts,
                                                           const uint32_t *__restrict__ offsets, uint32_t D,
                                                           uint2 *__restrict__ pairs, SortedAux aux) {
    culled_pairs_body<MODE>(blockIdx.x, N, perm, rect, depth, grid, rec, cb, counts, offsets, D, pairs, aux);
}
template <int THREADS, bool PAIRS>
__global__ __launch_bounds__(THREADS) void radix_hist_kernel(uint32_t n, const uint32_t *__restrict__ keys,
                                                             const uint2 *__restrict__ pairs, int shift, uint32_t mask,
                                                             uint32_t nblocks, uint32_t *__restrict__ table) {
    radix2_hist_body<THREADS, RADIX_ITEMS, PAIRS>(blockIdx.x, n, keys, pairs, shift, mask, nblocks, table);
}
template <int ITEMS_PER_1024>
__global__ __launch_bounds__(1024) void tile_hist_kernel(uint32_t n, const uint2 *__restrict__ pairs, uint32_t mask,
                                                         uint32_t nblocks, uint32_t *__restrict__ table, uint32_t ntiles,
                                                         uint32_t *__restrict__ tile_counts) {
    extern __shared__ uint32_t s_tiles_dyn[];
    radix2_hist_tiles_body<1024, ITEMS_PER_1024>(n, pairs, mask, nblocks, table, ntiles, tile_counts, s_tiles_dyn);
}
__global__ __launch_bounds__(1024) void ranges_from_counts_kernel(uint32_t ntiles, uint32_t *__restrict__ tile_counts,
                                                                  int32_t *__restrict__ ranges) {
    ranges_from_counts_body(ntiles, tile_counts, ranges);
}
__global__ __launch_bounds__(SCAN_THREADS) void radix_scan_kernel(uint32_t nblocks, uint32_t *__restrict__ table,
                                                                  uint32_t *__restrict__ totals) {
    radix_scan_body(blockIdx.x, nblocks, table, totals);
}
template <int NBITS, int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(uint32_t n, RadixIO io, int shift, uint32_t mask,
                                                                uint32_t nblocks, const uint32_t *__restrict__ table,
                                                                const uint32_t *__restrict__ totals) {
    radix2_scatter_body<NBITS, THREADS, RADIX_ITEMS, MODE>(blockIdx.x, n, io, shift, mask, nblocks, table, totals);
}
__global__ __launch_bounds__(256) void ranges_kernel(uint32_t D, const uint32_t *__restrict__ tile_keys,
                                                     int32_t *__restrict__ ranges) {
    ranges_body(blockIdx.x, D, tile_keys, ranges);
}

// the reference's 64-bit sort key, for callers that want it: tile_id<<32 | u32(depth)
__global__ __launch_bounds__(256) void build_keys_kernel(uint32_t D, const uint32_t *__restrict__ tile_keys,
                                                         const int32_t *__restrict__ vals,
                                                         const int32_t *__restrict__ depth,
                                                         uint64_t *__restrict__ keys) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= D) return;
    keys[i] = ((uint64_t)tile_keys[i] << 32) | (uint64_t)(uint32_t)depth[vals[i]];
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

// one stable LSD pass over n elements on ctx->stream; `mode` says where they come from and go to (RadixMode)
// tile_counts != NULL (first pass of a tile sort, shift == 0): the histogram kernel also counts the pairs per tile id
template <int THREADS>
int radix_pass_t(gs_ctx *ctx, uint32_t n, const RadixIO &io, int mode, int shift, int nbits, uint32_t ntiles = 0,
                 uint32_t *tile_counts = nullptr) {
    constexpr int TILE = THREADS * RADIX_ITEMS;
    const uint32_t nblocks = (n + TILE - 1) / TILE;
    int rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->radix_table, ((size_t)RADIX_BINS * nblocks + RADIX_BINS) * 4))) return rc;
    uint32_t *totals = (uint32_t *)ctx->radix_table.p;
    uint32_t *table = totals + RADIX_BINS;
    const uint32_t mask = (1u << nbits) - 1u;
    if (tile_counts) {
        // one workgroup per CU (its LDS holds the whole tile histogram), blocks of this pass's scatter size
        const uint32_t grid = nblocks < 256u ? nblocks : 256u;
        static bool lds_allowed = false;     // above 64 KiB of dynamic LDS the kernel has to be told so once
        if (!lds_allowed) {
            GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_hist_kernel<THREADS * RADIX_ITEMS / 1024>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TILE_HIST_MAX * 4));
            lds_allowed = true;
        }
        hipLaunchKernelGGL((tile_hist_kernel<THREADS * RADIX_ITEMS / 1024>), dim3(grid), dim3(1024), (size_t)ntiles * 4,
                           ctx->stream, n, io.pairs_in, mask, nblocks, table, ntiles, tile_counts);
    } else if (mode == RADIX_SOA)
        hipLaunchKernelGGL((radix_hist_kernel<THREADS, false>), dim3(nblocks), dim3(THREADS), 0, ctx->stream, n, io.keys_in,
                           (const uint2 *)nullptr, shift, mask, nblocks, table);
    else
        hipLaunchKernelGGL((radix_hist_kernel<THREADS, true>), dim3(nblocks), dim3(THREADS), 0, ctx->stream, n,
                           (const uint32_t *)nullptr, io.pairs_in, shift, mask, nblocks, table);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(RADIX_BINS), dim3(SCAN_THREADS), 0, ctx->stream, nblocks, table, totals);
#define GS_SCATTER_M(NB, M)                                                                                         \
    hipLaunchKernelGGL((radix_scatter_kernel<NB, THREADS, M>), dim3(nblocks), dim3(THREADS), 0, ctx->stream, n, io, \
                       shift, mask, nblocks, (const uint32_t *)table, (const uint32_t *)totals)
#define GS_SCATTER(NB)                                                                     \
    case NB:                                                                               \
        if (mode == RADIX_SOA) GS_SCATTER_M(NB, RADIX_SOA);                                \
        else if (mode == RADIX_PAIRS) GS_SCATTER_M(NB, RADIX_PAIRS);                       \
        else GS_SCATTER_M(NB, RADIX_PAIRS_TO_SOA);                                         \
        break;
    switch (nbits) {
        GS_SCATTER(1) GS_SCATTER(2) GS_SCATTER(3) GS_SCATTER(4) GS_SCATTER(5) GS_SCATTER(6) GS_SCATTER(7) GS_SCATTER(8)
        default: return gs_set_error(GS_ERR_ARG, "radix_pass: %d bits per pass", nbits);
    }
#undef GS_SCATTER
#undef GS_SCATTER_M
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int radix_pass(gs_ctx *ctx, uint32_t n, const RadixIO &io, int mode, int shift, int nbits, uint32_t ntiles = 0,
               uint32_t *tile_counts = nullptr) {
    return n >= RADIX_BIG_FROM ? radix_pass_t<RADIX_THREADS_BIG>(ctx, n, io, mode, shift, nbits, ntiles, tile_counts)
                               : radix_pass_t<RADIX_THREADS>(ctx, n, io, mode, shift, nbits, ntiles, tile_counts);
}

RadixIO soa_io(const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out, int32_t *vals_out) {
    RadixIO io{};
    io.keys_in = keys_in; io.vals_in = vals_in; io.keys_out = keys_out; io.vals_out = vals_out;
    return io;
}

int bits_of(uint32_t v) {
    int b = 0;
    while (v) { b++; v >>= 1; }
    return b;
}

}  // namespace

int gs_radix_sort_pairs(gs_ctx *ctx, uint32_t n, const uint32_t *keys_in, const int32_t *vals_in, int bits,
                        uint32_t *keys[2], int32_t *vals[2], const uint32_t **k_out, const int32_t **v_out) {
    if (bits < 1) bits = 1;
    const int passes = (bits + 7) / 8;
    const int per_pass = (bits + passes - 1) / passes;
    const uint32_t *kin = keys_in;
    const int32_t *vin = vals_in;
    for (int p = 0; p < passes; p++) {
        const int shift = p * per_pass;
        const int nbits = bits - shift < per_pass ? bits - shift : per_pass;
        const int rc = radix_pass(ctx, n, soa_io(kin, vin, keys[p & 1], vals[p & 1]), RADIX_SOA, shift, nbits);
        if (rc) return rc;
        kin = keys[p & 1];
        vin = vals[p & 1];
    }
    *k_out = kin;
    *v_out = vin;
    return GS_OK;
}

namespace {

// K4a: gaussians into (u32(depth), id) order, the low `depth_bits` bits of the key (the bits max(depth) occupies)
int bin_depth_sort(gs_ctx *ctx, int32_t N, const int32_t *depths, int depth_bits, const int32_t **perm_out,
                   const uint32_t **keys_out = nullptr) {
    int rc;
    const size_t n4 = ((size_t)N * 4 + 255) & ~(size_t)255;
    if ((rc = gs_buf_reserve(ctx, &ctx->depth_sort, 4 * n4))) return rc;
    char *ds = (char *)ctx->depth_sort.p;
    uint32_t *dk[2] = {(uint32_t *)ds, (uint32_t *)(ds + n4)};
    int32_t *dv[2] = {(int32_t *)(ds + 2 * n4), (int32_t *)(ds + 3 * n4)};
    const int depth_passes = (depth_bits + 7) / 8 > 0 ? (depth_bits + 7) / 8 : 1;
    gs_prof_scope prof_(ctx, GS_K_DEPTH_SORT);
    const uint32_t *kin = reinterpret_cast<const uint32_t *>(depths);
    const int32_t *vin = nullptr;
    for (int p = 0; p < depth_passes; p++) {
        const bool last = p == depth_passes - 1;
        const int nbits = last ? (depth_bits - 8 * p > 0 ? depth_bits - 8 * p : 1) : 8;
        if ((rc = radix_pass(ctx, (uint32_t)N, soa_io(kin, vin, last && !keys_out ? nullptr : dk[p & 1], dv[p & 1]), RADIX_SOA, 8 * p,
                             nbits)))
            return rc;
        kin = dk[p & 1];
        vin = dv[p & 1];
    }
    *perm_out = vin;
    if (keys_out) *keys_out = kin;      // the keys in sorted order (the last pass wrote them too)
    return GS_OK;
}

SortedAux sorted_aux(gs_ctx *ctx, int32_t N) {
    char *base = (char *)ctx->sorted_aux.p;
    const size_t n16 = (size_t)N * 16;
    return SortedAux{(int4 *)base, (float4 *)(base + n16), (float4 *)(base + 2 * n16), (uint2 *)(base + 3 * n16), nullptr};
}

// The tile count in its sorted form: depth sort over the low `bits` key bits, then the count in that order (counts[] and
// the per-gaussian summaries in depth order), per-block partials and the total.  Keys and ids travel together, so a
// sort over too few bits still counts correctly (D, max depth): only the order, and with it the lists, would be wrong.
int sorted_count_launch(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths, Grid grid,
                        const float4 *rec, int32_t tile_size, float pixel_offset, int bits) {
    int rc;
    ulonglong2 *partials = (ulonglong2 *)((char *)ctx->total.p + 16);
    const int32_t *perm = nullptr;
    const uint32_t *depth_sorted = nullptr;
    if ((rc = bin_depth_sort(ctx, N, depths, bits, &perm, &depth_sorted))) return rc;
    ctx->sorted_bits = bits;
    if ((rc = gs_buf_reserve(ctx, &ctx->sorted_aux, (size_t)N * 56))) return rc;
    SortedAux aux = sorted_aux(ctx, N);
    aux.depth_sorted = depth_sorted;
    // records written by K1 (gs_ctx_use_records) carry the tile rect in their fourth quarter: one gathered line
    const int4 *rect_src = rec && rec == ctx->ext_records ? nullptr : reinterpret_cast<const int4 *>(tiles_touched);
    gs_prof_scope prof_(ctx, GS_K_COUNT);
    if (rec) {
        const CullBox cb{(float)tile_size, pixel_offset, ctx->cull_shrink ? 1 : 0};
        hipLaunchKernelGGL(culled_pairs_kernel<PAIRS_COUNT_SORTED>, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                           perm, rect_src, depths, grid, rec, cb,
                           (uint32_t *)ctx->counts.p, (const uint32_t *)nullptr, 0u, (uint2 *)nullptr, aux);
        // counts and keys are both in sorted order: D and the largest key among the gaussians that own a pair
        hipLaunchKernelGGL(count_sum_kernel, dim3(COUNT_BLOCKS), dim3(SCAN_THREADS), 0, ctx->stream, N,
                           (const uint32_t *)ctx->counts.p, reinterpret_cast<const int32_t *>(depth_sorted), partials);
    } else {
        hipLaunchKernelGGL(count_sorted_kernel, dim3(COUNT_BLOCKS), dim3(SCAN_THREADS), 0, ctx->stream, N, perm,
                           reinterpret_cast<const int4 *>(tiles_touched), depths, grid, (uint32_t *)ctx->counts.p,
                           aux.info, partials, depth_sorted);
    }
    hipLaunchKernelGGL(count_total_kernel, dim3(1), dim3(COUNT_BLOCKS), 0, ctx->stream,
                       (const ulonglong2 *)partials, (unsigned long long *)ctx->total.p);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// rec == nullptr: the reference's rect semantics; otherwise pairs are culled with the box test.
// Launches the count and the copy of (D, max depth) to pinned memory, and records count_event.
// From ctx->sorted_count_from gaussians on (option sorted_count_from; SortedAux in binning_kernels.h) the depth
// sort runs here, over all 32 key bits (max(depth) is not known yet), and the count walks the gaussians in that order.
int bin_count_begin(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths, int32_t res_w,
                    int32_t res_h, int32_t tile_size, const float4 *rec, float pixel_offset) {
    int rc;
    if (ctx->prepare_pending_views > 0)
        return gs_set_error(GS_ERR_STATE, "gs_bin_count: a gs_bin_prepare_begin is pending on this ctx (its workspace is in use)");
    ctx->bin_culled = rec != nullptr;
    ctx->packed_fresh = false;    // only a gs_blend_forward that directly follows gs_bin_fill_culled may reuse the records
    ctx->bin_N = -1;
    ctx->bin_D = -1;
    ctx->bin_max_depth = 0;
    ctx->count_pending_N = -1;
    if (N == 0) {
        ctx->count_pending_N = 0;
        return GS_OK;
    }
    GS_HIP(hipSetDevice(ctx->device));
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    if ((rc = gs_buf_reserve(ctx, &ctx->counts, (size_t)N * 4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->total, 16 + COUNT_BLOCKS * sizeof(ulonglong2)))) return rc;
    if (!ctx->count_event) GS_HIP(hipEventCreateWithFlags(&ctx->count_event, hipEventDisableTiming));
    ulonglong2 *partials = (ulonglong2 *)((char *)ctx->total.p + 16);
    ctx->count_sorted = N >= ctx->sorted_count_from;
    if (ctx->count_sorted) {
        // max(depth) is not known yet: sort the key bytes the last view counted on this ctx needed (all four the
        // first time); bin_count_end checks the guess and bin_fill repeats sort + count with all bits if it was short
        const int bits = ctx->depth_bits_hint > 0 ? (ctx->depth_bits_hint + 7) / 8 * 8 : 32;
        if ((rc = sorted_count_launch(ctx, N, tiles_touched, depths, grid, rec, tile_size, pixel_offset, bits))) return rc;
    } else {
        if ((rc = gs_buf_reserve(ctx, &ctx->sorted_aux, (size_t)N * (rec ? 24 : 16)))) return rc;
        gs_prof_scope prof_(ctx, GS_K_COUNT);
        if (rec) {
            const CullBox cb{(float)tile_size, pixel_offset, ctx->cull_shrink ? 1 : 0};
            // (in this form the buffer holds the walked rect and the test mask of every gaussian, by gaussian id)
            hipLaunchKernelGGL(culled_pairs_kernel<PAIRS_COUNT>, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                               (const int32_t *)nullptr, reinterpret_cast<const int4 *>(tiles_touched), depths, grid,
                               rec, cb, (uint32_t *)ctx->counts.p, (const uint32_t *)nullptr, 0u, (uint2 *)nullptr,
                               SortedAux{(int4 *)ctx->sorted_aux.p, nullptr, nullptr, (uint2 *)((char *)ctx->sorted_aux.p + (size_t)N * 16), nullptr});
            hipLaunchKernelGGL(count_sum_kernel, dim3(COUNT_BLOCKS), dim3(SCAN_THREADS), 0, ctx->stream, N,
                               (const uint32_t *)ctx->counts.p, depths, partials);
        } else {
            hipLaunchKernelGGL(count_kernel, dim3(COUNT_BLOCKS), dim3(SCAN_THREADS), 0, ctx->stream, N,
                               reinterpret_cast<const int4 *>(tiles_touched), depths, grid,
                               (uint32_t *)ctx->counts.p, partials, (int4 *)ctx->sorted_aux.p);
        }
        hipLaunchKernelGGL(count_total_kernel, dim3(1), dim3(COUNT_BLOCKS), 0, ctx->stream,
                           (const ulonglong2 *)partials, (unsigned long long *)ctx->total.p);
    }
    GS_HIP(hipGetLastError());
    GS_HIP(hipMemcpyAsync(ctx->total_host, ctx->total.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipEventRecord(ctx->count_event, ctx->stream));
    ctx->count_pending_N = N;
    return GS_OK;
}

// waits for the count launched by bin_count_begin (only for it: later work on the stream may still run)
int bin_count_end(gs_ctx *ctx, int64_t *D_host) {
    GS_REQUIRE(D_host, "gs_bin_count: D_host is NULL");
    *D_host = 0;
    if (ctx->count_pending_N < 0)
        return gs_set_error(GS_ERR_STATE, "gs_bin_count_end: no gs_bin_count*_begin is pending on this ctx");
    const int32_t N = ctx->count_pending_N;
    ctx->count_pending_N = -1;
    if (N == 0) {
        ctx->bin_N = 0;
        ctx->bin_D = 0;
        return GS_OK;
    }
    if (const int rc = gs_wait_event(ctx, ctx->count_event, "gs_bin_count_end")) return rc;
    const uint64_t D = ctx->total_host[0];
    if (D >= (1ull << 31))
        return gs_set_error(GS_ERR_ARG, "gs_bin_count: %llu duplicates exceed the 2^31-1 limit",
                            (unsigned long long)D);
    ctx->bin_N = N;
    ctx->bin_D = (int64_t)D;
    ctx->bin_max_depth = (uint32_t)ctx->total_host[1];
    if (ctx->count_sorted) {      // was the guessed key width enough?  (bin_fill repeats sort + count if not)
        const int need = bits_of(ctx->bin_max_depth) > 0 ? bits_of(ctx->bin_max_depth) : 1;
        ctx->depth_bits_hint = need;
        ctx->sorted_redo = need > ctx->sorted_bits;
    }
    *D_host = (int64_t)D;
    return GS_OK;
}

// K3 + K4b + K5 of one view: pairs emitted in depth order (perm) at the scanned offsets, a stable sort by tile id
// alone, tile ranges (every tile's entry is written: tiles without pairs get (0, 0)).  D > 0.
// perm == nullptr: the gaussians' summaries in depth order are in ctx->sorted_aux (bin_count_begin's sorted form)
int bin_emit_sort(gs_ctx *ctx, int32_t N, const int4 *rect, const int32_t *depths, Grid grid, const float4 *rec,
                  int32_t tile_size, float pixel_offset, const int32_t *perm, const uint32_t *offsets, int64_t D,
                  uint64_t *keys, int32_t *vals, int32_t *ranges, bool own_count = false) {
    int rc;
    const int tiles = grid.tiles_x * grid.tiles_y;
    const size_t d4 = ((size_t)D * 4 + 255) & ~(size_t)255;
    int tile_bits = bits_of((uint32_t)(tiles - 1));
    if (tile_bits == 0) tile_bits = 1;
    const int tile_passes = (tile_bits + 7) / 8;
    const int bits_per_pass = (tile_bits + tile_passes - 1) / tile_passes;
    // pairs ping-pong (2 x D x 8 bytes); the last pass writes the caller's vals and the sorted tile ids for K5
    const size_t d8 = 2 * d4;
    if ((rc = gs_buf_reserve(ctx, &ctx->keys_tmp, 2 * d8))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->vals_tmp, d4))) return rc;
    uint2 *pp[2] = {(uint2 *)ctx->keys_tmp.p, (uint2 *)((char *)ctx->keys_tmp.p + d8)};
    uint32_t *sorted_tiles = (uint32_t *)ctx->vals_tmp.p;
    // K5 from a per-tile histogram taken in the first pass, when it fits one workgroup's LDS and nobody wants the
    // sorted tile ids (the 64-bit keys are built from them): the last pass then writes the gaussian ids only
    const bool hist_ranges = !keys && tiles <= TILE_HIST_MAX && ctx->tile_hist;
    uint32_t *tile_counts = nullptr;
    if (hist_ranges) {
        if ((rc = gs_buf_reserve(ctx, &ctx->tile_counts, (size_t)tiles * 4))) return rc;
        tile_counts = (uint32_t *)ctx->tile_counts.p;
        // ranges_from_counts leaves the counters it read at zero: a memset only for a new buffer or another frame
        if (ctx->tile_counts_clean != ctx->tile_counts.p || ctx->tile_counts_tiles < tiles)
            GS_HIP(hipMemsetAsync(tile_counts, 0, ctx->tile_counts.cap, ctx->stream));
        ctx->tile_counts_clean = nullptr;      // dirty until the ranges kernel below has been enqueued
    } else {
        // ranges_kernel only writes the tiles that own pairs (ranges_from_counts writes every tile)
        GS_HIP(hipMemsetAsync(ranges, 0, (size_t)tiles * 8, ctx->stream));
    }
    {
        gs_prof_scope prof_(ctx, GS_K_DUPLICATE);
        // own_count: walked rects and test masks of this ctx's own unsorted culled count (by gaussian id)
        const SortedAux aux = !perm ? sorted_aux(ctx, N)
                              : own_count ? SortedAux{(int4 *)ctx->sorted_aux.p, nullptr, nullptr,
                                                      (uint2 *)((char *)ctx->sorted_aux.p + (size_t)N * 16), nullptr}
                                          : SortedAux{};
        if (rec) {
            const CullBox cb{(float)tile_size, pixel_offset, ctx->cull_shrink ? 1 : 0};
            if (perm)
                hipLaunchKernelGGL(culled_pairs_kernel<PAIRS_EMIT>, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, perm,
                                   rect, depths, grid, rec, cb, (uint32_t *)nullptr, offsets, (uint32_t)D, pp[0], aux);
            else
                hipLaunchKernelGGL(culled_pairs_kernel<PAIRS_EMIT_AUX>, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                                   perm, rect, depths, grid, rec, cb, (uint32_t *)nullptr, offsets, (uint32_t)D, pp[0], aux);
        } else {
            // (perm with own_count: the clipped rects of this ctx's own count, by gaussian id)
            hipLaunchKernelGGL(duplicate_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, perm, rect,
                               depths, grid, offsets, (uint32_t)D, pp[0], (const int4 *)aux.info, perm ? 1 : 0);
        }
    }
    GS_HIP(hipGetLastError());
    {
        gs_prof_scope prof_(ctx, GS_K_SORT);
        for (int p = 0; p < tile_passes; p++) {
            const int shift = p * bits_per_pass;
            const int nbits = (tile_bits - shift < bits_per_pass) ? tile_bits - shift : bits_per_pass;
            const bool last = p == tile_passes - 1;
            RadixIO io{};
            io.pairs_in = pp[p & 1];
            io.pairs_out = pp[(p & 1) ^ 1];
            io.keys_out = hist_ranges ? nullptr : sorted_tiles;
            io.vals_out = vals;
            if ((rc = radix_pass(ctx, (uint32_t)D, io, last ? RADIX_PAIRS_TO_SOA : RADIX_PAIRS, shift, nbits, (uint32_t)tiles,
                                 p == 0 ? tile_counts : nullptr)))
                return rc;
        }
    }

    if (hist_ranges) {
        gs_prof_scope prof_(ctx, GS_K_RANGES);
        hipLaunchKernelGGL(ranges_from_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream, (uint32_t)tiles, tile_counts,
                           ranges);
        GS_HIP(hipGetLastError());
        ctx->tile_counts_clean = ctx->tile_counts.p;
        ctx->tile_counts_tiles = tiles;
        return GS_OK;
    }
    {
        gs_prof_scope prof_(ctx, GS_K_RANGES);
        hipLaunchKernelGGL(ranges_kernel, dim3(gs_div_up((int)D, 256)), dim3(256), 0, ctx->stream, (uint32_t)D,
                           (const uint32_t *)sorted_tiles, ranges);
        if (keys)
            hipLaunchKernelGGL(build_keys_kernel, dim3(gs_div_up((int)D, 256)), dim3(256), 0, ctx->stream, (uint32_t)D,
                               (const uint32_t *)sorted_tiles, (const int32_t *)vals, depths, keys);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int bin_fill(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths, int32_t res_w,
             int32_t res_h, int32_t tile_size, const float4 *rec, float pixel_offset, int64_t D, uint64_t *keys,
             int32_t *vals, int32_t *ranges) {
    int rc;
    GS_REQUIRE(ranges, "gs_bin_fill: ranges is NULL");
    ctx->packed_fresh = false;
    if (ctx->bin_N != N || ctx->bin_D != D || ctx->bin_culled != (rec != nullptr))
        return gs_set_error(GS_ERR_STATE, "gs_bin_fill: call gs_bin_count first with the same inputs "
                            "(have N=%d D=%lld, got N=%d D=%lld)", ctx->bin_N, (long long)ctx->bin_D, N,
                            (long long)D);
    GS_HIP(hipSetDevice(ctx->device));
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    const int tiles = grid.tiles_x * grid.tiles_y;
    if (D == 0) {
        GS_HIP(hipMemsetAsync(ranges, 0, (size_t)tiles * 8, ctx->stream));
        return GS_OK;
    }
    GS_REQUIRE(vals, "gs_bin_fill: vals is NULL with D=%lld", (long long)D);
    const int4 *rect = reinterpret_cast<const int4 *>(tiles_touched);
    // K4a: gaussians into (u32(depth), id) order; only the bytes max(depth) occupies are sorted.  (Done already, and
    // the counts are in that order, when the count ran in its sorted form: perm stays NULL for the kernels below.)
    const int32_t *perm = nullptr;
    if (!ctx->count_sorted) {
        if ((rc = bin_depth_sort(ctx, N, depths, bits_of(ctx->bin_max_depth), &perm))) return rc;
    } else if (ctx->sorted_redo) {     // the count sorted fewer key bits than this view's depths occupy
        if ((rc = sorted_count_launch(ctx, N, tiles_touched, depths, grid, rec, tile_size, pixel_offset, 32))) return rc;
        ctx->sorted_redo = false;
    }

    // K2': exclusive offsets of tiles-per-gaussian in depth order
    const int nblocks = gs_div_up(N, SCAN_TILE);
    if ((rc = gs_buf_reserve(ctx, &ctx->offsets, (size_t)N * 4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->block_sums, (size_t)nblocks * 4))) return rc;
    const uint32_t *counts = (const uint32_t *)ctx->counts.p;
    uint32_t *bs = (uint32_t *)ctx->block_sums.p;
    {
        gs_prof_scope prof_(ctx, GS_K_SCAN);
        // the first kernel leaves the counts in depth order in the offsets array, the third turns them into offsets
        // in place (one gather of counts[perm[i]] instead of two)
        uint32_t *offs = (uint32_t *)ctx->offsets.p;
        hipLaunchKernelGGL(sorted_block_sums_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, N, perm,
                           counts, bs, perm ? offs : (uint32_t *)nullptr);
        // up to 2048 blocks (2 M gaussians) every block of the third kernel sums the totals in front of it itself
        const bool raw = nblocks <= 2048;
        if (!raw) hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, nblocks, bs);
        hipLaunchKernelGGL(sorted_offsets_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, N,
                           (const int32_t *)nullptr, perm ? (const uint32_t *)offs : counts, (const uint32_t *)bs, offs,
                           raw ? 1 : 0);
    }
    GS_HIP(hipGetLastError());

    return bin_emit_sort(ctx, N, rect, depths, grid, rec, tile_size, pixel_offset, perm, (const uint32_t *)ctx->offsets.p, D,
                         keys, vals, ranges, perm != nullptr);
}

int cull_records(gs_ctx *ctx, int32_t N, const float *centers, const float *colors, const float *opacities,
                 const float *conics, bool reuse, const float4 **rec) {
    GS_HIP(hipSetDevice(ctx->device));
    return gs_pack_records(ctx, N, centers, colors, opacities, conics, reuse, rec);
}

}  // namespace

// Workspace of the per-view entry points for up to N_max gaussians and D_max list entries, allocated once.
extern "C" int gs_ctx_reserve(gs_ctx *ctx, int32_t N_max, int64_t D_max, int32_t res_w, int32_t res_h,
                              int32_t tile_size, int32_t array_api, int32_t batch_views) {
    GS_REQUIRE(ctx, "gs_ctx_reserve: ctx is NULL");
    GS_REQUIRE(N_max >= 0 && D_max >= 0 && D_max < (1ll << 31), "gs_ctx_reserve: N_max=%d D_max=%lld out of range", N_max,
               (long long)D_max);
    GS_REQUIRE(res_w > 0 && res_h > 0 && tile_size > 0, "gs_ctx_reserve: invalid frame %dx%d tile %d", res_w, res_h,
               tile_size);
    GS_REQUIRE(batch_views >= 0 && batch_views <= GS_BIN_MAX_VIEWS, "gs_ctx_reserve: batch_views=%d not in [0, %d]",
               batch_views, GS_BIN_MAX_VIEWS);
    GS_HIP(hipSetDevice(ctx->device));
    const size_t nv = batch_views > 1 ? (size_t)batch_views : 1;      // gs_bin_prepare_*: one slice per view
    const size_t N = (size_t)(N_max > 0 ? N_max : 1), D = (size_t)(D_max > 0 ? D_max : 1);
    const size_t n4 = (N * 4 + 255) & ~(size_t)255, d4 = (D * 4 + 255) & ~(size_t)255;
    const size_t small_tile = (size_t)RADIX_THREADS * RADIX_ITEMS;
    const size_t radix_blocks = ((N > D ? N : D) + small_tile - 1) / small_tile;       // the small tile bounds both
    const size_t blocks16 = (size_t)gs_div_up(res_w, 16) * gs_div_up(res_h, 16);
    struct { gs_buf *b; size_t bytes; } want[] = {
        {&ctx->counts, nv * n4}, {&ctx->total, 16 + COUNT_BLOCKS * sizeof(ulonglong2)}, {&ctx->depth_sort, nv * 4 * n4},
        {&ctx->offsets, N * 4}, {&ctx->block_sums, nv * (((size_t)gs_div_up((int)N, SCAN_TILE) * 4 + 255) & ~(size_t)255)},
        {&ctx->keys_tmp, 4 * d4}, {&ctx->vals_tmp, d4},
        {&ctx->radix_table, nv * ((((size_t)RADIX_BINS * (radix_blocks + 1) + RADIX_BINS) * 4 + 255) & ~(size_t)255)},
        {&ctx->bounds, batch_views ? (size_t)GS_BIN_MAX_VIEWS * 256 * sizeof(ulonglong2) + 2 * GS_BIN_MAX_VIEWS * 8 : 0},
        {&ctx->tile_order_buf, blocks16 * 2 * sizeof(int32_t)},
        {&ctx->tile_counts, (size_t)gs_div_up(res_w, tile_size) * gs_div_up(res_h, tile_size) * 4}, {&ctx->loss_partials, 1024 * sizeof(double)},
        {&ctx->sorted_aux, N_max >= ctx->sorted_count_from ? N * 56 : N * 24},
        {&ctx->packed, array_api ? N * 64 : 0}, {&ctx->grad_rows, array_api ? N * 64 : 0}};
    for (auto &w : want)
        if (w.bytes)
            if (const int rc = gs_buf_reserve(ctx, w.b, w.bytes, true)) return rc;
    if (!ctx->count_event) GS_HIP(hipEventCreateWithFlags(&ctx->count_event, hipEventDisableTiming));
    if (batch_views && !ctx->bounds_event) GS_HIP(hipEventCreateWithFlags(&ctx->bounds_event, hipEventDisableTiming));
    return GS_OK;
}

extern "C" int gs_bin_count(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                            int32_t res_w, int32_t res_h, int32_t tile_size, int64_t *D_host) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    GS_REQUIRE(D_host, "gs_bin_count: D_host is NULL");
    if ((rc = bin_count_begin(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, nullptr, 0.0f))) return rc;
    return bin_count_end(ctx, D_host);
}

extern "C" int gs_bin_count_begin(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                                  int32_t res_w, int32_t res_h, int32_t tile_size) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    return bin_count_begin(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, nullptr, 0.0f);
}

extern "C" int gs_bin_fill(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                           int32_t res_w, int32_t res_h, int32_t tile_size, int64_t D, uint64_t *keys,
                           int32_t *vals, int32_t *ranges) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    return bin_fill(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, nullptr, 0.0f, D, keys, vals, ranges);
}

extern "C" int gs_bin_fill_prepared(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                                    const float *records, const int32_t *perm, const uint32_t *offsets, int64_t D,
                                    int32_t res_w, int32_t res_h, int32_t tile_size, float pixel_offset, uint64_t *keys,
                                    int32_t *vals, int32_t *ranges) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    GS_REQUIRE(ranges, "gs_bin_fill_prepared: ranges is NULL");
    GS_REQUIRE(D >= 0 && D < (1ll << 31), "gs_bin_fill_prepared: D=%lld out of range", (long long)D);
    GS_REQUIRE(((uintptr_t)records & 15) == 0, "gs_bin_fill_prepared: records must be 16-byte aligned");
    ctx->packed_fresh = false;
    GS_HIP(hipSetDevice(ctx->device));
    const Grid grid{gs_div_up(res_w, tile_size), gs_div_up(res_h, tile_size)};
    if (D == 0 || N == 0) {
        GS_HIP(hipMemsetAsync(ranges, 0, (size_t)grid.tiles_x * grid.tiles_y * 8, ctx->stream));
        return GS_OK;
    }
    GS_REQUIRE(vals && perm && offsets, "gs_bin_fill_prepared: NULL vals / perm / offsets with D=%lld", (long long)D);
    return bin_emit_sort(ctx, N, reinterpret_cast<const int4 *>(tiles_touched), depths, grid,
                         reinterpret_cast<const float4 *>(records), tile_size, pixel_offset, perm, offsets, D, keys, vals,
                         ranges);
}

extern "C" int gs_bin_count_culled_begin(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched,
                                         const int32_t *depths, const float *centers, const float *colors,
                                         const float *opacities, const float *conics, int32_t res_w,
                                         int32_t res_h, int32_t tile_size, float pixel_offset) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    const float4 *rec = nullptr;
    if ((rc = cull_records(ctx, N, centers, colors, opacities, conics, false, &rec))) return rc;
    return bin_count_begin(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, rec, pixel_offset);
}

extern "C" int gs_bin_count_end(gs_ctx *ctx, int64_t *D_host) {
    GS_REQUIRE(ctx, "gs_bin_count_end: ctx is NULL");
    return bin_count_end(ctx, D_host);
}

extern "C" int gs_bin_count_culled(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                                   const float *centers, const float *colors, const float *opacities,
                                   const float *conics, int32_t res_w, int32_t res_h, int32_t tile_size,
                                   float pixel_offset, int64_t *D_host) {
    GS_REQUIRE(D_host, "gs_bin_count_culled: D_host is NULL");
    int rc = gs_bin_count_culled_begin(ctx, N, tiles_touched, depths, centers, colors, opacities, conics, res_w,
                                       res_h, tile_size, pixel_offset);
    if (rc) return rc;
    return bin_count_end(ctx, D_host);
}

extern "C" int gs_bin_fill_culled(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                                  const float *centers, const float *colors, const float *opacities,
                                  const float *conics, int32_t res_w, int32_t res_h, int32_t tile_size,
                                  float pixel_offset, int64_t D, uint64_t *keys, int32_t *vals, int32_t *ranges) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    const float4 *rec = nullptr;
    // the records packed by gs_bin_count_culled from the same arrays are still current
    if ((rc = cull_records(ctx, N, centers, colors, opacities, conics, true, &rec))) return rc;
    rc = bin_fill(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, rec, pixel_offset, D, keys, vals, ranges);
    ctx->packed_fresh = rc == GS_OK;
    return rc;
}
