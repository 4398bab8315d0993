// Device code of the binning kernels (K2-K5), shared by the per-view entry points (binning.hip) and the batched,
// host-wait-free ones (bin_batch.hip): every kernel body is a function of its block index and plain pointers,
// and the __global__ wrappers of the two files only differ in where those come from.
#pragma once
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // elements per block

constexpr int RADIX_BINS = 256;
constexpr int RADIX_ITEMS = 16;                              // elements per thread of a radix pass
constexpr int RADIX_THREADS = 256;                           // small sorts: 4096 elements per workgroup
constexpr int RADIX_THREADS_BIG = 512;                       // large sorts: 8192 (fewer, longer digit runs)
constexpr uint32_t RADIX_BIG_FROM = 32u << 20;               // elements from which the large tile is used

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
// Windows that lie inside ONE run -- most windows of a rect of hundreds of tiles -- take a shortcut: the run open at
// the end of the previous window (`carry`) is asked for its end with a readlane, and if it covers this window too every
// slot belongs to it (*uniform = true) with no LDS traffic at all.
__device__ __forceinline__ uint32_t slot_owner(uint32_t *mark_row, int lane, uint32_t cnt, uint32_t first,
                                               uint32_t sbase, uint32_t &carry, bool *uniform = nullptr) {
    if (uniform) *uniform = false;
    if (carry != 0u) {
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)(first + cnt), (int)(carry - 1u));
        if (end >= sbase + 64u) {
            if (uniform) *uniform = true;
            return carry - 1u;
        }
    }
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

// K2'a: per-block totals of counts taken in depth order (perm == nullptr: the counts are in that order already)
// (counts_sorted, if given, receives the counts in depth order, so that K2'c streams them instead of gathering again;
// it may be K2'c's own output array: that kernel reads an element before the same thread overwrites it)
__device__ __forceinline__ void sorted_block_sums_body(int bid, int N, const int32_t *__restrict__ perm,
                                                                         const uint32_t *__restrict__ counts,
                                                                         uint32_t *__restrict__ block_sums,
                                                                         uint32_t *__restrict__ counts_sorted = nullptr) {
    const int base = bid * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < N) {
            const uint32_t c = counts[perm ? perm[base + k] : base + k];
            if (counts_sorted) counts_sorted[base + k] = c;
            sum += c;
        }
    uint32_t total;
    block_exclusive_scan(sum, &total);
    if (threadIdx.x == 0) block_sums[bid] = total;
}

// K2'b: single block turns block_sums into exclusive block offsets
// (*total_out, if given, receives the grand total: the number of pairs D of the view)
__device__ __forceinline__ void scan_blocks_body(int bid, int nblocks, uint32_t *__restrict__ block_sums,
                                                 uint32_t *__restrict__ total_out) {
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
    if (total_out && threadIdx.x == 0) *total_out = carry_s;     // behind the loop's last barrier
}

// K2'c: exclusive offsets per sorted position
__device__ __forceinline__ void sorted_offsets_body(int bid, int N, const int32_t *__restrict__ perm,
                                                                      const uint32_t *counts,
                                                                      const uint32_t *__restrict__ block_offsets,
                                                                      uint32_t *offsets, bool raw_sums = false) {
    const int base = bid * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        c[k] = (base + k < N) ? counts[perm ? perm[base + k] : base + k] : 0u;
        sum += c[k];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, &total);
    if (raw_sums) {
        // block_offsets holds the per-block TOTALS: this block adds up the ones in front of it itself (a few KiB from
        // L2) and the one-workgroup scan kernel between K2'a and K2'c is not launched
        uint32_t before = 0;
        for (int b = threadIdx.x; b < bid; b += SCAN_THREADS) before += block_offsets[b];
        uint32_t all;
        block_exclusive_scan(before, &all);
        run += all;
    } else {
        run += block_offsets[bid];
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < N) offsets[base + k] = run;
        run += c[k];
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
    int shrink;        // 1: walk only the tiles inside the ellipse's own bounding box (cull_shrink below)
};

// The rect the count and emit passes walk is the reference's bounding square of 3 sigma of the LARGER axis; the pairs
// that can pass the box test lie inside the axis-aligned bounding box of the ellipse Q(u, v) <= qmax (qmax from the
// test's own threshold), which is smaller for elongated or faint gaussians.  Shrinking the walked rect to that box,
// grown by 0.1 % and one pixel, only skips pairs the exact test rejects (a tile wholly beyond u_max has
// Q >= u^2 det / gamma > qmax by that margin): the lists do not change, fewer tests run.  Conics that are not
// positive definite keep their rect (the test keeps every pair of theirs).
__device__ __forceinline__ void cull_shrink(const float4 geo, const float4 con, CullBox cb, int &x0, int &y0, int &w, int &h) {
    const float al = -con.x, be = -con.y, ga = -con.z;
    if (!(al > 0.0f && ga > 0.0f && be * be < 4.0f * al * ga)) return;
    if (geo.w >= 1e29f) { w = 0; h = 0; return; }            // the test rejects every pair
    const float qmax = (1e-4f - geo.w) / 0.999f;             // the test passes iff qmin * 0.999 <= 1e-4 - tau
    if (qmax < 0.0f) { w = 0; h = 0; return; }               // qmin >= 0: nothing passes
    const float det = al * ga - 0.25f * be * be;
    if (!(det > 0.0f) || !(qmax < 1e30f)) return;
    const float um = sqrtf(qmax * ga / det) * 1.001f + 1.0f, vm = sqrtf(qmax * al / det) * 1.001f + 1.0f;
    if (!(um < 1e9f) || !(vm < 1e9f)) return;
    // tile t covers pixels [t * tile - off, t * tile - off + tile - 1]; it can pass only if that meets [c - m, c + m]
    const float lim = 1e9f;
    const float fx0 = fminf(fmaxf((geo.x - um + cb.off - (cb.tile - 1.0f)) / cb.tile, -lim), lim);
    const float fx1 = fminf(fmaxf((geo.x + um + cb.off) / cb.tile, -lim), lim);
    const float fy0 = fminf(fmaxf((geo.y - vm + cb.off - (cb.tile - 1.0f)) / cb.tile, -lim), lim);
    const float fy1 = fminf(fmaxf((geo.y + vm + cb.off) / cb.tile, -lim), lim);
    if (!(fx0 == fx0) || !(fx1 == fx1) || !(fy0 == fy0) || !(fy1 == fy1)) return;      // NaN centre: keep
    const int nx0 = max(x0, (int)ceilf(fx0)), nx1 = min(x0 + w - 1, (int)floorf(fx1));
    const int ny0 = max(y0, (int)ceilf(fy0)), ny1 = min(y0 + h - 1, (int)floorf(fy1));
    if (nx1 < nx0 || ny1 < ny0) { w = 0; h = 0; return; }
    x0 = nx0; y0 = ny0; w = nx1 - nx0 + 1; h = ny1 - ny0 + 1;
}

__device__ __forceinline__ bool pair_may_contribute(const float4 geo, const float4 con, const float2 halves, int tx, int ty,
                                                    CullBox cb) {
    const float xlo = (float)tx * cb.tile - cb.off, ylo = (float)ty * cb.tile - cb.off;
    return gs_box_may_contribute(geo, con, xlo, xlo + cb.tile - 1.0f, ylo, ylo + cb.tile - 1.0f, halves);
}

// One wave per 64 consecutive gaussians; the wave walks the concatenation of their rects 64 pairs at a
// time (slot_owner finds each pair's gaussian, as in K3), so a gaussian covering thousands of tiles does not
// serialise its wave.  CULL_EMIT = false: counts[g] <- pairs that pass.  CULL_EMIT = true (gaussians taken
// in depth order through perm): the passing pairs are written as {tile id, gaussian} from offsets[first position
// of the wave] on, compacted with a ballot, i.e. in (depth order, tile order).
//
// Large scenes (SortedAux, modes 2 and 3): once rect / depth / records of all gaussians no longer fit the caches, the
// gathers in depth order cost a 128-byte line per 16-byte item, twice (count + scan read counts[perm[i]], emit reads
// rect / depth / record of perm[i]).  There the depth sort runs FIRST and the count walks the gaussians in depth
// order: it gathers once, leaves counts[] in that order (the scans stream it) and a 48-byte summary per gaussian
// (clipped rect, id, the two record quarters of the box test) that the emit pass streams instead of gathering again.
//
// Box test once (mask != nullptr, any mode): a gaussian whose clipped rect has at most 64 tiles -- nearly all of them --
// gets the outcome of its tests as a 64-bit mask from the count pass (bit = position in the rect), and the emit pass
// reads the bit instead of evaluating the quadratic again (and does not load the record quarters of such a gaussian);
// larger rects are tested in both passes as before.  mask[] is indexed like counts[] (gaussian id in modes 0 / 1,
// depth-order position in modes 2 / 3).
struct SortedAux {
    int4 *info;      // x0 | w << 16, y0, h, gaussian
    float4 *geo;     // record quarter 0 (culled lists only)
    float4 *con;     // record quarter 1
    uint2 *mask;     // test outcomes of rects of <= 64 tiles
    const uint32_t *depth_sorted;   // mode 2: the depth keys in depth order (the sort's own output) -- no gather
};

enum { PAIRS_COUNT = 0, PAIRS_EMIT = 1, PAIRS_COUNT_SORTED = 2, PAIRS_EMIT_AUX = 3 };

template <int MODE>
__device__ __forceinline__ void culled_pairs_body(int bid, int N, const int32_t *__restrict__ perm,
                                                           const int4 *__restrict__ rect,
                                                           const int32_t *__restrict__ depth, Grid grid,
                                                           const float4 *__restrict__ rec, CullBox cb,
                                                           uint32_t *__restrict__ counts,
                                                           const uint32_t *__restrict__ offsets, uint32_t D,
                                                           uint2 *__restrict__ pairs, SortedAux aux = SortedAux{}) {
    constexpr bool CULL_EMIT = MODE == PAIRS_EMIT || MODE == PAIRS_EMIT_AUX;
    __shared__ uint32_t s_first[4][64];  // exclusive prefix of rect sizes within the wave
    __shared__ int4 s_info[4][64];       // x0, y0, h, gaussian
    __shared__ float4 s_geo[4][64], s_con[4][64];
    __shared__ float2 s_half[4][64];     // the two divisions of the box test, once per gaussian
    __shared__ uint32_t s_pass[4][64];
    __shared__ uint32_t s_mark[4][64];
    __shared__ uint32_t s_cnt[4][64], s_mlo[4][64], s_mhi[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (bid * 4 + wave) * 64;
    const int i = i0 + lane;
    const bool use_mask = aux.mask != nullptr;
    uint32_t cnt = 0;
    int x0 = 0, y0 = 0, h = 1, g = 0;
    float4 geo = make_float4(0.f, 0.f, 0.f, 0.f), con = geo;
    uint2 bits = make_uint2(0u, 0u);
    if (MODE == PAIRS_EMIT_AUX) {
        if (i < N) {
            const int4 inf = aux.info[i];
            x0 = inf.x & 0xFFFF; y0 = inf.y; h = inf.z; g = inf.w;
            cnt = (uint32_t)(inf.x >> 16) * (uint32_t)h;
            if (cnt && use_mask && cnt <= 64u) {
                bits = aux.mask[i];
            } else if (cnt) {
                geo = aux.geo[i];
                con = aux.con[i];
            }
        }
    } else if (MODE == PAIRS_EMIT && aux.info != nullptr) {
        // this ctx's own count pass left the walked rect (and the mask) of every gaussian, by gaussian id: two small
        // gathers instead of rect + depth + record
        if (i < N) {
            g = perm[i];
            const int4 inf = aux.info[g];
            x0 = inf.x & 0xFFFF; y0 = inf.y; h = inf.z;
            cnt = (uint32_t)(inf.x >> 16) * (uint32_t)h;
            if (cnt && use_mask && cnt <= 64u) {
                bits = aux.mask[g];
            } else if (cnt) {
                geo = rec[(size_t)g * 4];
                con = rec[(size_t)g * 4 + 1];
            }
        }
    } else if (i < N) {
        g = MODE == PAIRS_COUNT ? i : perm[i];
        // perm may be the caller's array (gs_bin_fill_prepared): an id that is not a gaussian owns no pair
        if ((uint32_t)g < (uint32_t)N) {
            if (MODE == PAIRS_COUNT_SORTED) {
                // one gathered line per gaussian where K1 wrote the records (their fourth quarter is the tile rect)
                const float4 rq = rect ? make_float4(0.f, 0.f, 0.f, 0.f) : rec[(size_t)g * 4 + 3];
                const int4 r = rect ? rect[g] : make_int4(__float_as_int(rq.x), __float_as_int(rq.y), __float_as_int(rq.z),
                                                          __float_as_int(rq.w));
                cnt = tile_count(r, aux.depth_sorted ? (int)aux.depth_sorted[i] : depth[g], grid, x0, y0, h);
            } else {
                cnt = tile_count(rect[g], depth[g], grid, x0, y0, h);
            }
        } else {
            g = 0;
        }
        if (cnt && cb.shrink) {        // (the record is read first: its line is needed below in any case)
            const float4 g0 = rec[(size_t)g * 4], c0 = rec[(size_t)g * 4 + 1];
            int w = (int)(cnt / (uint32_t)h);
            cull_shrink(g0, c0, cb, x0, y0, w, h);
            cnt = (uint32_t)w * (uint32_t)h;
            if (cnt == 0u) h = 1;
        }
        if (cnt && MODE == PAIRS_EMIT && use_mask && cnt <= 64u) {
            bits = aux.mask[g];
        } else if (cnt) {
            geo = rec[(size_t)g * 4];
            con = rec[(size_t)g * 4 + 1];
        }
        if (MODE == PAIRS_COUNT_SORTED || (MODE == PAIRS_COUNT && aux.info != nullptr)) {
            const uint32_t w = cnt ? cnt / (uint32_t)h : 0u;      // cnt = w * h; no pair: w = 0
            aux.info[i] = make_int4(x0 | (int)(w << 16), y0, h, g);
        }
        if (MODE == PAIRS_COUNT_SORTED) {
            aux.geo[i] = geo;
            aux.con[i] = con;
        }
    }
    // Count passes with masks: a gaussian whose walked rect has at most SOLO tiles -- the typical one -- is tested by
    // its own lane in a short loop (tiles in the rect's order: ty fastest), with no owner search, LDS atomics or
    // per-window synchronisation; only the larger rects go through the cooperative walk below.
    constexpr uint32_t SOLO = 16;
    const uint32_t cnt_all = cnt;
    uint32_t solo_bits = 0;
    bool solo = false;
    if (!CULL_EMIT && use_mask) {
        solo = cnt > 0u && cnt <= SOLO;
        const float2 hv = gs_box_halves(con);
        int tx = x0, ty = y0;
        for (uint32_t k = 0; k < SOLO; k++) {
            const bool on = solo && k < cnt;
            if (__ballot(on) == 0ull) break;
            if (on && pair_may_contribute(geo, con, hv, tx, ty, cb)) solo_bits |= 1u << k;
            if (++ty == y0 + h) { ty = y0; tx++; }
        }
        if (solo) cnt = 0u;           // nothing left for the cooperative walk
    }
    const uint32_t inc = wave_inclusive_scan(cnt);
    const uint32_t first = inc - cnt;
    s_first[wave][lane] = first;
    s_info[wave][lane] = make_int4(x0, y0, h, g);
    s_geo[wave][lane] = geo;
    s_con[wave][lane] = con;
    s_half[wave][lane] = gs_box_halves(con);
    s_pass[wave][lane] = 0u;
    s_cnt[wave][lane] = cnt;
    s_mlo[wave][lane] = bits.x;
    s_mhi[wave][lane] = bits.y;
    const uint32_t wave_total = __shfl(inc, 63, 64);   // 0 when the whole wave is past N
    uint32_t out = (CULL_EMIT && i0 < N) ? offsets[i0] : 0u;
    uint32_t carry = 0;
    for (uint32_t sbase = 0; sbase < wave_total; sbase += 64) {      // wave-uniform trip count
        const uint32_t sidx = sbase + lane;
        bool one_owner;
        const int lo = (int)slot_owner(s_mark[wave], lane, cnt, first, sbase, carry, &one_owner);
        if (one_owner && use_mask && s_cnt[wave][lo] <= 64u) one_owner = false;     // a run of exactly 64: it has a mask
        bool pass = false;
        int tx = 0, ty = 0;
        if (sidx < wave_total) {
            const int4 inf = s_info[wave][lo];
            const uint32_t local = sidx - s_first[wave][lo];
            tx = inf.x + (int)(local / (uint32_t)inf.z);
            ty = inf.y + (int)(local % (uint32_t)inf.z);
            if (CULL_EMIT && use_mask && s_cnt[wave][lo] <= 64u)      // decided by the count pass
                pass = (((local < 32u ? s_mlo[wave][lo] : s_mhi[wave][lo]) >> (local & 31u)) & 1u) != 0u;
            else
                pass = pair_may_contribute(s_geo[wave][lo], s_con[wave][lo], s_half[wave][lo], tx, ty, cb);
            // (a window inside one run of more than 64 tiles is counted with a ballot below: 64 LDS atomics on one
            // address are 64 serial operations)
            if (!CULL_EMIT && pass && !one_owner) {
                if (use_mask && s_cnt[wave][lo] <= 64u) atomicOr(local < 32u ? &s_mlo[wave][lo] : &s_mhi[wave][lo], 1u << (local & 31u));
                else atomicAdd(&s_pass[wave][lo], 1u);
            }
        }
        if (!CULL_EMIT && one_owner) {       // the run covers the whole window, so it has more than 64 tiles: no mask
            const uint32_t n_pass = (uint32_t)__popcll(__ballot(pass));
            if (lane == 0) s_pass[wave][lo] += n_pass;
        }
        if (CULL_EMIT) {
            const unsigned long long m = __ballot(pass);
            if (pass) {
                const uint32_t pos = out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < D)       // always true: the count pass ran the same test on the same data
                    pairs[pos] = make_uint2((uint32_t)(tx * grid.tiles_y + ty), (uint32_t)s_info[wave][lo].w);
            }
            out += (uint32_t)__popcll(m);
        }
    }
    if (!CULL_EMIT) {
        __syncthreads();
        if (i < N) {
            const bool small = use_mask && cnt_all <= 64u;
            const uint32_t lo32 = solo ? solo_bits : s_mlo[wave][lane], hi32 = solo ? 0u : s_mhi[wave][lane];
            counts[i] = small ? (uint32_t)(__popc(lo32) + __popc(hi32)) : s_pass[wave][lane];
            if (use_mask) aux.mask[i] = make_uint2(lo32, hi32);      // (mode 0: i is the gaussian id)
        }
    }
}

// K5: per-tile [start,end) from the sorted tile ids
__device__ __forceinline__ void ranges_body(int bid, uint32_t D, const uint32_t *__restrict__ tile_keys,
                                                     int32_t *__restrict__ ranges) {
    const uint32_t i = bid * 256u + threadIdx.x;
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

}  // namespace
