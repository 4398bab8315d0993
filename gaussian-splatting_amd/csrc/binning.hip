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
// The radix sort is hand-written for wave64: per pass one histogram kernel (LDS atomics), one scan
// kernel (one workgroup per digit turns the [digit][block] table into global offsets) and one
// scatter kernel that ranks 4096 keys per workgroup with ballot matching (stable inside a wave,
// per-wave counters in LDS across waves), reorders them through LDS and writes each digit run
// contiguously.
#include "common.h"
#include <cstring>

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // elements per block

constexpr int RADIX_THREADS = 256;
constexpr int RADIX_WAVES = RADIX_THREADS / 64;
constexpr int RADIX_ITEMS = 16;
constexpr int RADIX_TILE = RADIX_THREADS * RADIX_ITEMS;      // 4096 keys per workgroup
constexpr int RADIX_WAVE_TILE = 64 * RADIX_ITEMS;            // 1024 keys per wave
constexpr int RADIX_BINS = 256;

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

// Wave64 inclusive scans on the DPP crossbar: row_shr 1, 2, 4, 8 inside each row of 16 lanes, then the last lane of
// row 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31).  Lanes without a source
// take the identity.  Six VALU steps; a __shfl_up ladder is six dependent trips through the LDS crossbar.
#define GS_DPP_SCAN(OP, IDENT)                                                                        \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x111, 0xf, 0xf, false));   \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x112, 0xf, 0xf, false));   \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x114, 0xf, 0xf, false));   \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x118, 0xf, 0xf, false));   \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x142, 0xa, 0xf, false));   \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, 0x143, 0xc, 0xf, false));
__device__ __forceinline__ uint32_t gs_add_u32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t gs_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ inline uint32_t wave_inclusive_scan(uint32_t v) {
    GS_DPP_SCAN(gs_add_u32, 0u)
    return v;
}
// running maximum (identity 0)
__device__ inline uint32_t wave_inclusive_max(uint32_t v) {
    GS_DPP_SCAN(gs_max_u32, 0u)
    return v;
}
#undef GS_DPP_SCAN

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

// K2: tiles-per-gaussian; per-block partial sums and the largest u32(depth) among the gaussians
// that emit anything (it decides how many depth-sort passes are needed).  COUNT_BLOCKS blocks
// grid-stride over N; count_total_kernel folds the partials (no same-address atomics).
constexpr int COUNT_BLOCKS = 1024;

__global__ __launch_bounds__(SCAN_THREADS) void count_kernel(int N, const int4 *__restrict__ rect,
                                                             const int32_t *__restrict__ depth, Grid grid,
                                                             uint32_t *__restrict__ counts,
                                                             ulonglong2 *__restrict__ partials) {
    __shared__ uint32_t s_max[SCAN_THREADS / 64];
    __shared__ unsigned long long s_sum[SCAN_THREADS / 64];
    unsigned long long c = 0;
    uint32_t md = 0;
    for (int g = blockIdx.x * SCAN_THREADS + threadIdx.x; g < N; g += COUNT_BLOCKS * SCAN_THREADS) {
        int x0, y0, h;
        const int d = depth[g];
        const uint32_t cg = tile_count(rect[g], d, grid, x0, y0, h);
        counts[g] = cg;
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

// K2'a: per-block totals of counts taken in depth order
__global__ __launch_bounds__(SCAN_THREADS) void sorted_block_sums_kernel(int N, const int32_t *__restrict__ perm,
                                                                         const uint32_t *__restrict__ counts,
                                                                         uint32_t *__restrict__ block_sums) {
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < N) sum += counts[perm[base + k]];
    uint32_t total;
    block_exclusive_scan(sum, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// K2'b: single block turns block_sums into exclusive block offsets
__global__ __launch_bounds__(1024) void scan_blocks_kernel(int nblocks, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(v);
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, chunk_total = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) wbase += wave_sums[w];
            chunk_total += wave_sums[w];
        }
        const uint32_t carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + wbase + inc - v;   // D < 2^31 is enforced on the host
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + chunk_total;
        __syncthreads();
    }
}

// K2'c: exclusive offsets per sorted position
__global__ __launch_bounds__(SCAN_THREADS) void sorted_offsets_kernel(int N, const int32_t *__restrict__ perm,
                                                                      const uint32_t *__restrict__ counts,
                                                                      const uint32_t *__restrict__ block_offsets,
                                                                      uint32_t *__restrict__ offsets) {
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        c[k] = (base + k < N) ? counts[perm[base + k]] : 0u;
        sum += c[k];
    }
    uint32_t total;
    uint32_t run = block_offsets[blockIdx.x] + block_exclusive_scan(sum, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < N) offsets[base + k] = run;
        run += c[k];
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The 64 lanes of a wave hold runs of `cnt` consecutive output slots starting at `first` (its exclusive prefix).
// Returns, for slot sbase + lane of the concatenation, the lane that owns it: every lane whose run starts inside
// the window [sbase, sbase + 64) marks its first slot in the wave's LDS row, and a running maximum carries the
// owner over the rest of its run (owners grow with the slot); slots before the window's first mark belong to the
// run that was open at the end of the previous window (`carry`, wave-uniform, 0 at the start).  One LDS round
// trip and six DPP steps per window -- a binary search over the prefix array is six dependent LDS reads, and the
// kernels that walk rects are bound by exactly that latency.  LDS operations of one wave execute in order.
__device__ __forceinline__ uint32_t slot_owner(uint32_t *mark_row, int lane, uint32_t cnt, uint32_t first,
                                               uint32_t sbase, uint32_t &carry) {
    mark_row[lane] = 0u;
    wave_lds_sync();
    if (cnt != 0u && first >= sbase && first < sbase + 64u) mark_row[first - sbase] = (uint32_t)lane + 1u;
    wave_lds_sync();
    uint32_t m = wave_inclusive_max(mark_row[lane]);
    m = max(m, carry);
    carry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
    wave_lds_sync();                  // the row is rewritten by the next window
    return m - 1u;                    // a slot below the wave's total always has an owner (m >= 1)
}

// K3: duplicate.  One wave per 64 consecutive positions of the depth order; the wave's output
// slots are a contiguous range, filled cooperatively (64 slots per step, slot_owner) so stores are
// coalesced and a gaussian covering hundreds of tiles does not serialise its wave.
__global__ __launch_bounds__(256) void duplicate_kernel(int N, const int32_t *__restrict__ perm,
                                                        const int4 *__restrict__ rect,
                                                        const int32_t *__restrict__ depth, Grid grid,
                                                        const uint32_t *__restrict__ offsets,
                                                        uint32_t *__restrict__ tile_keys, int32_t *__restrict__ vals) {
    __shared__ uint32_t s_first[4][64];  // exclusive prefix of counts within the wave
    __shared__ int4 s_info[4][64];       // x0, y0, h, gaussian
    __shared__ uint32_t s_mark[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wave) * 64;
    const int i = i0 + lane;
    uint32_t cnt = 0;
    int x0 = 0, y0 = 0, h = 1, g = 0;
    if (i < N) {
        g = perm[i];
        cnt = tile_count(rect[g], depth[g], grid, x0, y0, h);
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
        if (sidx < wave_total) {
            const int4 inf = s_info[wave][lo];
            const uint32_t local = sidx - s_first[wave][lo];
            const int tx = inf.x + (int)(local / (uint32_t)inf.z);
            const int ty = inf.y + (int)(local % (uint32_t)inf.z);
            tile_keys[out_base + sidx] = (uint32_t)(tx * grid.tiles_y + ty);
            vals[out_base + sidx] = inf.w;
        }
    }
}

// ------------------------------------------------------------------ culled variants of K2 / K3
// The reference's rect (Q8/Q9) is the bounding square of 3 sigma of the LARGER axis: about half of its
// tiles are never reached by the ellipse at alpha >= 1/255.  With the packed blend records at hand
// the same exact box test the blend kernels use when staging decides per (gaussian, tile) pair, so
// pairs that cannot change any pixel are never emitted, sorted or staged.  A list built this way is
// the reference's list with those entries removed (same order); images and gradients are unchanged.
struct CullBox {
    float tile, off;   // tile size in pixels, pixel_offset
};

__device__ __forceinline__ bool pair_may_contribute(const float4 geo, const float4 con, int tx, int ty, CullBox cb) {
    const float xlo = (float)tx * cb.tile - cb.off, ylo = (float)ty * cb.tile - cb.off;
    return gs_box_may_contribute(geo, con, xlo, xlo + cb.tile - 1.0f, ylo, ylo + cb.tile - 1.0f);
}

// One wave per 64 consecutive gaussians; the wave walks the concatenation of their rects 64 pairs at a
// time (slot_owner finds each pair's gaussian, as in K3), so a gaussian covering thousands of tiles does not
// serialise its wave.  CULL_EMIT = false: counts[g] <- pairs that pass.  CULL_EMIT = true (gaussians taken
// in depth order through perm): the passing pairs are written from offsets[first position of the wave]
// on, compacted with a ballot, i.e. in (depth order, tile order).
template <bool CULL_EMIT>
__global__ __launch_bounds__(256) void culled_pairs_kernel(int N, const int32_t *__restrict__ perm,
                                                           const int4 *__restrict__ rect,
                                                           const int32_t *__restrict__ depth, Grid grid,
                                                           const float4 *__restrict__ rec, CullBox cb,
                                                           uint32_t *__restrict__ counts,
                                                           const uint32_t *__restrict__ offsets, uint32_t D,
                                                           uint32_t *__restrict__ tile_keys,
                                                           int32_t *__restrict__ vals) {
    __shared__ uint32_t s_first[4][64];  // exclusive prefix of rect sizes within the wave
    __shared__ int4 s_info[4][64];       // x0, y0, h, gaussian
    __shared__ float4 s_geo[4][64], s_con[4][64];
    __shared__ uint32_t s_pass[4][64];
    __shared__ uint32_t s_mark[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wave) * 64;
    const int i = i0 + lane;
    uint32_t cnt = 0;
    int x0 = 0, y0 = 0, h = 1, g = 0;
    float4 geo = make_float4(0.f, 0.f, 0.f, 0.f), con = geo;
    if (i < N) {
        g = CULL_EMIT ? perm[i] : i;
        cnt = tile_count(rect[g], depth[g], grid, x0, y0, h);
        if (cnt) {
            geo = rec[(size_t)g * 4];
            con = rec[(size_t)g * 4 + 1];
        }
    }
    const uint32_t inc = wave_inclusive_scan(cnt);
    const uint32_t first = inc - cnt;
    s_first[wave][lane] = first;
    s_info[wave][lane] = make_int4(x0, y0, h, g);
    s_geo[wave][lane] = geo;
    s_con[wave][lane] = con;
    s_pass[wave][lane] = 0u;
    const uint32_t wave_total = __shfl(inc, 63, 64);   // 0 when the whole wave is past N
    uint32_t out = (CULL_EMIT && i0 < N) ? offsets[i0] : 0u;
    uint32_t carry = 0;
    for (uint32_t sbase = 0; sbase < wave_total; sbase += 64) {      // wave-uniform trip count
        const uint32_t sidx = sbase + lane;
        const int lo = (int)slot_owner(s_mark[wave], lane, cnt, first, sbase, carry);
        bool pass = false;
        int tx = 0, ty = 0;
        if (sidx < wave_total) {
            const int4 inf = s_info[wave][lo];
            const uint32_t local = sidx - s_first[wave][lo];
            tx = inf.x + (int)(local / (uint32_t)inf.z);
            ty = inf.y + (int)(local % (uint32_t)inf.z);
            pass = pair_may_contribute(s_geo[wave][lo], s_con[wave][lo], tx, ty, cb);
        }
        if (CULL_EMIT) {
            const unsigned long long m = __ballot(pass);
            if (pass) {
                const uint32_t pos = out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < D) {     // always true: the count pass ran the same test on the same data
                    tile_keys[pos] = (uint32_t)(tx * grid.tiles_y + ty);
                    vals[pos] = s_info[wave][lo].w;
                }
            }
            out += (uint32_t)__popcll(m);
        } else if (pass) {
            atomicAdd(&s_pass[wave][lo], 1u);
        }
    }
    if (!CULL_EMIT) {
        __syncthreads();
        if (i < N) counts[i] = s_pass[wave][lane];
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

// ---------------------------------------------------------------------------- radix sort pass
// digit = (key >> shift) & mask, at most 8 bits.  Block b owns keys [b*4096, (b+1)*4096).

__global__ __launch_bounds__(RADIX_THREADS) void radix_hist_kernel(uint32_t n, const uint32_t *__restrict__ keys,
                                                                   int shift, uint32_t mask, uint32_t nblocks,
                                                                   uint32_t *__restrict__ table) {
    __shared__ uint32_t h[RADIX_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)RADIX_TILE;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + base);   // base is a multiple of 4096
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS / 4; k++) {
        const uint32_t e = (k * RADIX_THREADS + threadIdx.x) * 4u;
        const uint32_t i = base + e;
        if (i + 3 < n) {
            const uint4 v = k4[k * RADIX_THREADS + threadIdx.x];
            atomicAdd(&h[(v.x >> shift) & mask], 1u);
            atomicAdd(&h[(v.y >> shift) & mask], 1u);
            atomicAdd(&h[(v.z >> shift) & mask], 1u);
            atomicAdd(&h[(v.w >> shift) & mask], 1u);
        } else {
            for (uint32_t j = i; j < n && j < i + 4; j++) atomicAdd(&h[(keys[j] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    table[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: table[d][b] <- keys with digit d in blocks before b; totals[d] <- keys
// with digit d.  (No atomics: thousands of same-address atomics serialise at ~12 ns each.)
__global__ __launch_bounds__(RADIX_THREADS) void radix_scan_kernel(uint32_t nblocks, uint32_t *__restrict__ table,
                                                                   uint32_t *__restrict__ totals) {
    const uint32_t d = blockIdx.x;
    uint32_t *row = table + (size_t)d * nblocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += RADIX_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(v, &total);
        if (i < nblocks) row[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) totals[d] = carry;
}

// vals_in == nullptr: the value of key i is i (first pass of the depth sort).
// keys_out == nullptr: only the values are wanted (last pass of the depth sort).
// NBITS (digit width) is a template parameter: the ballot matching below is the kernel's cost (it is bound
// by instruction issue, not by HBM), and a runtime width puts a branch around every one of its steps.
template <int NBITS>
__global__ __launch_bounds__(RADIX_THREADS) void radix_scatter_kernel(
    uint32_t n, const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out, int shift, uint32_t mask,
    uint32_t nblocks, const uint32_t *__restrict__ table, const uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_cnt[RADIX_WAVES][RADIX_BINS];   // per-wave digit counts -> exclusive wave bases
    __shared__ uint32_t s_lbase[RADIX_BINS];              // where digit d starts inside this block
    __shared__ uint32_t s_gbase[RADIX_BINS];              // global start of (d, block) minus s_lbase[d]
    __shared__ uint32_t s_keys[RADIX_TILE];
    __shared__ int32_t s_vals[RADIX_TILE];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)RADIX_TILE;
    const uint32_t wbase = base + (uint32_t)w * RADIX_WAVE_TILE;
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (int c = t; c < RADIX_WAVES * RADIX_BINS; c += RADIX_THREADS) (&s_cnt[0][0])[c] = 0u;

    // wave-striped arrangement: the block's key order is (wave, item, lane)
    uint32_t key[RADIX_ITEMS], rank[RADIX_ITEMS];
    int32_t val[RADIX_ITEMS];
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; k++) {
        const uint32_t i = wbase + k * 64 + lane;
        key[k] = 0xFFFFFFFFu;
        val[k] = 0;
        if (i < n) {
            key[k] = keys_in[i];
            val[k] = vals_in ? vals_in[i] : (int32_t)i;
        }
    }
    __syncthreads();

    // plain LDS accesses: a volatile pointer here decays to a generic one and every counter access becomes a
    // flat load / store with a full wait.  LDS operations of one wave execute in order; the fences keep the
    // compiler from moving them across each other.
    uint32_t *cnt = &s_cnt[w][0];
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; k++) {
        const bool valid = wbase + k * 64 + lane < n;
        const uint32_t d = (key[k] >> shift) & mask;
        // lanes of this wave holding the same digit: per bit, keep the lanes whose bit equals mine.
        // ~(ballot ^ m) with m = all ones where my bit is set: one xnor + one and per half and bit.
        const unsigned long long vb = __ballot(valid);
        uint32_t peers_lo = (uint32_t)vb, peers_hi = (uint32_t)(vb >> 32);
#pragma unroll
        for (int b = 0; b < NBITS; b++) {
            const uint32_t bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit != 0u);
            const uint32_t m = 0u - bit;
            peers_lo &= ~((uint32_t)bal ^ m);
            peers_hi &= ~((uint32_t)(bal >> 32) ^ m);
        }
        const unsigned long long peers = ((unsigned long long)peers_hi << 32) | peers_lo;
        const uint32_t r = __popc(peers_lo & (uint32_t)lt) + __popc(peers_hi & (uint32_t)(lt >> 32));
        const uint32_t prev = cnt[d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && r == 0) cnt[d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rank[k] = prev + r;     // position among this wave's keys of digit d
    }
    __syncthreads();

    {   // thread t = digit t
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < RADIX_WAVES; ww++) {
            const uint32_t c = s_cnt[ww][t];
            s_cnt[ww][t] = run;
            run += c;
        }
        uint32_t block_total, all;
        const uint32_t excl = block_exclusive_scan(run, &block_total);
        const uint32_t dbase = block_exclusive_scan(totals[t], &all);   // keys with a smaller digit
        s_lbase[t] = excl;
        s_gbase[t] = dbase + table[(size_t)t * nblocks + blockIdx.x] - excl;
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; k++) {
        if (wbase + k * 64 + lane < n) {
            const uint32_t d = (key[k] >> shift) & mask;
            const uint32_t pos = s_lbase[d] + s_cnt[w][d] + rank[k];
            s_keys[pos] = key[k];
            s_vals[pos] = val[k];
        }
    }
    __syncthreads();

    const uint32_t count = min((uint32_t)RADIX_TILE, n - base);
    for (uint32_t i = t; i < count; i += RADIX_THREADS) {
        const uint32_t kk = s_keys[i];
        const uint32_t pos = s_gbase[(kk >> shift) & mask] + i;
        if (keys_out) keys_out[pos] = kk;
        vals_out[pos] = s_vals[i];
    }
}

// K5: per-tile [start,end) from the sorted tile ids
__global__ __launch_bounds__(256) void ranges_kernel(uint32_t D, const uint32_t *__restrict__ tile_keys,
                                                     int32_t *__restrict__ ranges) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= D) return;
    const uint32_t t = tile_keys[i];
    if (i == 0) {
        ranges[2 * t] = 0;
    } else {
        const uint32_t p = tile_keys[i - 1];
        if (p != t) {
            ranges[2 * p + 1] = (int32_t)i;
            ranges[2 * t] = (int32_t)i;
        }
    }
    if (i == D - 1) ranges[2 * t + 1] = (int32_t)D;
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

// one stable LSD pass over n (key, value) pairs on ctx->stream
int radix_pass(gs_ctx *ctx, uint32_t n, const uint32_t *keys_in, const int32_t *vals_in, uint32_t *keys_out,
               int32_t *vals_out, int shift, int nbits) {
    const uint32_t nblocks = (n + RADIX_TILE - 1) / RADIX_TILE;
    int rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->radix_table, ((size_t)RADIX_BINS * nblocks + RADIX_BINS) * 4))) return rc;
    uint32_t *totals = (uint32_t *)ctx->radix_table.p;
    uint32_t *table = totals + RADIX_BINS;
    const uint32_t mask = (1u << nbits) - 1u;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblocks), dim3(RADIX_THREADS), 0, ctx->stream, n, keys_in, shift,
                       mask, nblocks, table);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(RADIX_BINS), dim3(RADIX_THREADS), 0, ctx->stream, nblocks, table,
                       totals);
#define GS_SCATTER(NB)                                                                                          \
    case NB:                                                                                                    \
        hipLaunchKernelGGL(radix_scatter_kernel<NB>, dim3(nblocks), dim3(RADIX_THREADS), 0, ctx->stream, n, keys_in, \
                           vals_in, keys_out, vals_out, shift, mask, nblocks, (const uint32_t *)table,          \
                           (const uint32_t *)totals);                                                           \
        break;
    switch (nbits) {
        GS_SCATTER(1) GS_SCATTER(2) GS_SCATTER(3) GS_SCATTER(4) GS_SCATTER(5) GS_SCATTER(6) GS_SCATTER(7) GS_SCATTER(8)
        default: return gs_set_error(GS_ERR_ARG, "radix_pass: %d bits per pass", nbits);
    }
#undef GS_SCATTER
    GS_HIP(hipGetLastError());
    return GS_OK;
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
        const int rc = radix_pass(ctx, n, kin, vin, keys[p & 1], vals[p & 1], shift, nbits);
        if (rc) return rc;
        kin = keys[p & 1];
        vin = vals[p & 1];
    }
    *k_out = kin;
    *v_out = vin;
    return GS_OK;
}

namespace {

// rec == nullptr: the reference's rect semantics; otherwise pairs are culled with the box test.
// Launches the count and the copy of (D, max depth) to pinned memory, and records count_event.
int bin_count_begin(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths, int32_t res_w,
                    int32_t res_h, int32_t tile_size, const float4 *rec, float pixel_offset) {
    int rc;
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
    {
        gs_prof_scope prof_(ctx, GS_K_COUNT);
        if (rec) {
            const CullBox cb{(float)tile_size, pixel_offset};
            hipLaunchKernelGGL(culled_pairs_kernel<false>, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                               (const int32_t *)nullptr, reinterpret_cast<const int4 *>(tiles_touched), depths, grid,
                               rec, cb, (uint32_t *)ctx->counts.p, (const uint32_t *)nullptr, 0u, (uint32_t *)nullptr,
                               (int32_t *)nullptr);
            hipLaunchKernelGGL(count_sum_kernel, dim3(COUNT_BLOCKS), dim3(SCAN_THREADS), 0, ctx->stream, N,
                               (const uint32_t *)ctx->counts.p, depths, partials);
        } else {
            hipLaunchKernelGGL(count_kernel, dim3(COUNT_BLOCKS), dim3(SCAN_THREADS), 0, ctx->stream, N,
                               reinterpret_cast<const int4 *>(tiles_touched), depths, grid,
                               (uint32_t *)ctx->counts.p, partials);
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
    GS_HIP(hipEventSynchronize(ctx->count_event));
    const uint64_t D = ctx->total_host[0];
    if (D >= (1ull << 31))
        return gs_set_error(GS_ERR_ARG, "gs_bin_count: %llu duplicates exceed the 2^31-1 limit",
                            (unsigned long long)D);
    ctx->bin_N = N;
    ctx->bin_D = (int64_t)D;
    ctx->bin_max_depth = (uint32_t)ctx->total_host[1];
    *D_host = (int64_t)D;
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
    GS_HIP(hipMemsetAsync(ranges, 0, (size_t)tiles * 8, ctx->stream));
    if (D == 0) return GS_OK;
    GS_REQUIRE(vals, "gs_bin_fill: vals is NULL with D=%lld", (long long)D);
    const int4 *rect = reinterpret_cast<const int4 *>(tiles_touched);
    const size_t n4 = ((size_t)N * 4 + 255) & ~(size_t)255, d4 = ((size_t)D * 4 + 255) & ~(size_t)255;

    // K4a: gaussians into (u32(depth), id) order; only the bytes max(depth) occupies are sorted
    if ((rc = gs_buf_reserve(ctx, &ctx->depth_sort, 4 * n4))) return rc;
    char *ds = (char *)ctx->depth_sort.p;
    uint32_t *dk[2] = {(uint32_t *)ds, (uint32_t *)(ds + n4)};
    int32_t *dv[2] = {(int32_t *)(ds + 2 * n4), (int32_t *)(ds + 3 * n4)};
    const int depth_bits = bits_of(ctx->bin_max_depth);
    const int depth_passes = (depth_bits + 7) / 8 > 0 ? (depth_bits + 7) / 8 : 1;
    const int32_t *perm = nullptr;
    {
        gs_prof_scope prof_(ctx, GS_K_DEPTH_SORT);
        const uint32_t *kin = reinterpret_cast<const uint32_t *>(depths);
        const int32_t *vin = nullptr;
        for (int p = 0; p < depth_passes; p++) {
            const bool last = p == depth_passes - 1;
            const int nbits = last ? (depth_bits - 8 * p > 0 ? depth_bits - 8 * p : 1) : 8;
            if ((rc = radix_pass(ctx, (uint32_t)N, kin, vin, last ? nullptr : dk[p & 1], dv[p & 1], 8 * p, nbits)))
                return rc;
            kin = dk[p & 1];
            vin = dv[p & 1];
        }
        perm = vin;
    }

    // K2': exclusive offsets of tiles-per-gaussian in depth order
    const int nblocks = gs_div_up(N, SCAN_TILE);
    if ((rc = gs_buf_reserve(ctx, &ctx->offsets, (size_t)N * 4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->block_sums, (size_t)nblocks * 4))) return rc;
    const uint32_t *counts = (const uint32_t *)ctx->counts.p;
    uint32_t *bs = (uint32_t *)ctx->block_sums.p;
    {
        gs_prof_scope prof_(ctx, GS_K_SCAN);
        hipLaunchKernelGGL(sorted_block_sums_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, N, perm,
                           counts, bs);
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, nblocks, bs);
        hipLaunchKernelGGL(sorted_offsets_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, ctx->stream, N, perm, counts,
                           (const uint32_t *)bs, (uint32_t *)ctx->offsets.p);
    }
    GS_HIP(hipGetLastError());

    // K3 + K4b: duplicate in depth order, then a stable sort by tile id alone
    int tile_bits = bits_of((uint32_t)(tiles - 1));
    if (tile_bits == 0) tile_bits = 1;
    const int tile_passes = (tile_bits + 7) / 8;
    const int bits_per_pass = (tile_bits + tile_passes - 1) / tile_passes;
    if ((rc = gs_buf_reserve(ctx, &ctx->keys_tmp, 2 * d4))) return rc;
    if ((rc = gs_buf_reserve(ctx, &ctx->vals_tmp, d4))) return rc;
    uint32_t *tk[2] = {(uint32_t *)ctx->keys_tmp.p, (uint32_t *)((char *)ctx->keys_tmp.p + d4)};
    // value buffers alternate so that the last pass writes the caller's array
    int32_t *vb[2] = {(int32_t *)ctx->vals_tmp.p, vals};
    int cur = (tile_passes & 1) ? 0 : 1;
    {
        gs_prof_scope prof_(ctx, GS_K_DUPLICATE);
        if (rec) {
            const CullBox cb{(float)tile_size, pixel_offset};
            hipLaunchKernelGGL(culled_pairs_kernel<true>, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, perm,
                               rect, depths, grid, rec, cb, (uint32_t *)nullptr, (const uint32_t *)ctx->offsets.p,
                               (uint32_t)D, tk[0], vb[cur]);
        } else {
            hipLaunchKernelGGL(duplicate_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, perm, rect,
                               depths, grid, (const uint32_t *)ctx->offsets.p, tk[0], vb[cur]);
        }
    }
    GS_HIP(hipGetLastError());
    int kcur = 0;
    {
        gs_prof_scope prof_(ctx, GS_K_SORT);
        for (int p = 0; p < tile_passes; p++) {
            const int shift = p * bits_per_pass;
            const int nbits = (tile_bits - shift < bits_per_pass) ? tile_bits - shift : bits_per_pass;
            if ((rc = radix_pass(ctx, (uint32_t)D, tk[kcur], vb[cur], tk[kcur ^ 1], vb[cur ^ 1], shift, nbits)))
                return rc;
            kcur ^= 1;
            cur ^= 1;
        }
    }
    // here vb[cur] == vals

    {
        gs_prof_scope prof_(ctx, GS_K_RANGES);
        hipLaunchKernelGGL(ranges_kernel, dim3(gs_div_up((int)D, 256)), dim3(256), 0, ctx->stream, (uint32_t)D,
                           (const uint32_t *)tk[kcur], ranges);
        if (keys)
            hipLaunchKernelGGL(build_keys_kernel, dim3(gs_div_up((int)D, 256)), dim3(256), 0, ctx->stream, (uint32_t)D,
                               (const uint32_t *)tk[kcur], (const int32_t *)vals, depths, keys);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int cull_records(gs_ctx *ctx, int32_t N, const float *centers, const float *colors, const float *opacities,
                 const float *conics, bool reuse, const float4 **rec) {
    GS_HIP(hipSetDevice(ctx->device));
    return gs_pack_records(ctx, N, centers, colors, opacities, conics, reuse, rec);
}

}  // namespace

extern "C" int gs_bin_count(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                            int32_t res_w, int32_t res_h, int32_t tile_size, int64_t *D_host) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    GS_REQUIRE(D_host, "gs_bin_count: D_host is NULL");
    if ((rc = bin_count_begin(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, nullptr, 0.0f))) return rc;
    return bin_count_end(ctx, D_host);
}

extern "C" int gs_bin_fill(gs_ctx *ctx, int32_t N, const int32_t *tiles_touched, const int32_t *depths,
                           int32_t res_w, int32_t res_h, int32_t tile_size, int64_t D, uint64_t *keys,
                           int32_t *vals, int32_t *ranges) {
    int rc = check_bin_args(ctx, N, tiles_touched, depths, res_w, res_h, tile_size);
    if (rc) return rc;
    return bin_fill(ctx, N, tiles_touched, depths, res_w, res_h, tile_size, nullptr, 0.0f, D, keys, vals, ranges);
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
