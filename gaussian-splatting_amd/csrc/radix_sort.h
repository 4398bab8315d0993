// Stable LSD radix pass for wave64, sized for the 160 KiB of LDS of a gfx950 CU (the sorts of K4: gaussians by
// depth key, (tile, gaussian) pairs by tile id, list positions by gaussian id in the ordered K7 reduction).
//
// A pass = histogram (one [digit][block] table) -> scan (one workgroup per digit) -> scatter.  A scatter
// workgroup ranks TILE = THREADS * ITEMS elements by ballot matching (stable inside a wave; per-wave LDS counters
// order the waves), reorders them through LDS and writes every digit run as one contiguous piece.  What sets its
// speed is the size of those pieces in HBM, not the ranking arithmetic: with 256 digits a 4096-key tile writes
// 64-byte pieces per array, a 16384-element tile of 8-byte (key, value) pairs writes 512-byte ones (a 256-byte
// contiguous piece per wave instruction is what plain stores need for their full rate, MI355X_MICROARCH.md).
// Hence the two knobs: the tile (template parameters, up to one 1024-thread workgroup holding all of a CU's LDS)
// and the element layout (MODE): separate key and value arrays, or one array of uint2 {key, value}.
//
// Blocks are numbered so that the blocks resident on one XCD at a time are NEIGHBOURS in the input: workgroups
// are dealt round-robin over the 8 XCDs, so launch slot s runs logical block (s % 8) * ceil(nblocks / 8) + s / 8
// and the adjacent pieces that neighbouring blocks write to a digit's run meet in ONE L2 instead of eight.
#pragma once
#include "binning_kernels.h"

namespace {

enum RadixMode {
    RADIX_SOA = 0,        // keys_in[n] (+ vals_in[n] or NULL = index) -> keys_out[n] (or NULL) + vals_out[n]
    RADIX_PAIRS = 1,      // pairs_in[n] {key, value} -> pairs_out[n]
    RADIX_PAIRS_TO_SOA = 2  // pairs_in[n] -> keys_out[n] (or NULL) + vals_out[n]   (last pass: the caller's array)
};

struct RadixIO {
    const uint32_t *keys_in;
    const int32_t *vals_in;
    const uint2 *pairs_in;
    uint32_t *keys_out;
    int32_t *vals_out;
    uint2 *pairs_out;
};

// launch slot -> logical block: consecutive logical blocks share an XCD (see the header comment)
__device__ __forceinline__ uint32_t radix_logical_block(uint32_t slot, uint32_t nblocks) {
#ifdef GS_RADIX_NO_XCD_REMAP
    return slot;
#endif
    const uint32_t per = nblocks >> 3, rem = nblocks & 7u, x = slot & 7u, j = slot >> 3;
    return x * per + min(x, rem) + j;
}

template <int THREADS, int ITEMS, bool PAIRS>
__device__ __forceinline__ void radix2_hist_body(uint32_t slot, uint32_t n, const uint32_t *__restrict__ keys,
                                                 const uint2 *__restrict__ pairs, int shift, uint32_t mask,
                                                 uint32_t nblocks, uint32_t *__restrict__ table) {
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t h[RADIX_BINS];
    for (int c = threadIdx.x; c < RADIX_BINS; c += THREADS) h[c] = 0;
    __syncthreads();
    const uint32_t bid = radix_logical_block(slot, nblocks);
    const uint32_t base = bid * (uint32_t)TILE;
    if (PAIRS) {
        const uint4 *p4 = reinterpret_cast<const uint4 *>(pairs + base);     // two pairs per 16-byte load
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            const uint32_t e = (k * THREADS + threadIdx.x) * 2u, i = base + e;
            if (i + 1 < n) {
                const uint4 v = p4[k * THREADS + threadIdx.x];
                atomicAdd(&h[(v.x >> shift) & mask], 1u);
                atomicAdd(&h[(v.z >> shift) & mask], 1u);
            } else if (i < n) {
                atomicAdd(&h[(pairs[i].x >> shift) & mask], 1u);
            }
        }
    } else {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys + base);
#pragma unroll
        for (int k = 0; k < ITEMS / 4; k++) {
            const uint32_t e = (k * THREADS + threadIdx.x) * 4u, i = base + e;
            if (i + 3 < n) {
                const uint4 v = k4[k * THREADS + threadIdx.x];
                atomicAdd(&h[(v.x >> shift) & mask], 1u);
                atomicAdd(&h[(v.y >> shift) & mask], 1u);
                atomicAdd(&h[(v.z >> shift) & mask], 1u);
                atomicAdd(&h[(v.w >> shift) & mask], 1u);
            } else {
                for (uint32_t j = i; j < n && j < i + 4; j++) atomicAdd(&h[(keys[j] >> shift) & mask], 1u);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < RADIX_BINS; c += THREADS) table[(size_t)c * nblocks + bid] = h[c];
}

// First pass of the tile sort, with the per-TILE histogram folded in: besides the [digit][block] table of the pass
// a workgroup counts its pairs per tile id in LDS (up to 32 768 tiles = 128 KiB: a workgroup may hold all of a CU's
// 160 KiB) and adds the counts to tile_counts[] once, at its end.  The exclusive scan of those counts IS the tile
// ranges (K5), so the last pass need not write the sorted tile ids and no kernel has to read them back: 8 bytes per
// pair less.  Persistent: `gridDim.x` workgroups walk the blocks with a stride; THREADS = 1024 keeps 16 waves on
// the CU that the one resident workgroup occupies.
template <int THREADS, int ITEMS>
__device__ __forceinline__ void radix2_hist_tiles_body(uint32_t n, const uint2 *__restrict__ pairs, uint32_t mask,
                                                       uint32_t nblocks, uint32_t *__restrict__ table, uint32_t ntiles,
                                                       uint32_t *__restrict__ tile_counts, uint32_t *s_tiles) {
    constexpr int TILE = THREADS * ITEMS;        // the block size of the pass's scatter kernel
    __shared__ uint32_t h[RADIX_BINS];
    for (uint32_t c = threadIdx.x; c < ntiles; c += THREADS) s_tiles[c] = 0u;
    for (uint32_t slot = blockIdx.x; slot < nblocks; slot += gridDim.x) {
        for (int c = threadIdx.x; c < RADIX_BINS; c += THREADS) h[c] = 0;
        __syncthreads();
        const uint32_t bid = radix_logical_block(slot, nblocks);
        const uint32_t base = bid * (uint32_t)TILE;
        const uint4 *p4 = reinterpret_cast<const uint4 *>(pairs + base);     // two pairs per 16-byte load
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            const uint32_t e = (k * THREADS + threadIdx.x) * 2u, i = base + e;
            if (i + 1 < n) {
                const uint4 v = p4[k * THREADS + threadIdx.x];
                atomicAdd(&h[v.x & mask], 1u);
                atomicAdd(&h[v.z & mask], 1u);
                if (v.x < ntiles) atomicAdd(&s_tiles[v.x], 1u);
                if (v.z < ntiles) atomicAdd(&s_tiles[v.z], 1u);
            } else if (i < n) {
                const uint32_t key = pairs[i].x;
                atomicAdd(&h[key & mask], 1u);
                if (key < ntiles) atomicAdd(&s_tiles[key], 1u);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < RADIX_BINS; c += THREADS) table[(size_t)c * nblocks + bid] = h[c];
        __syncthreads();
    }
    for (uint32_t c = threadIdx.x; c < ntiles; c += THREADS) {
        const uint32_t v = s_tiles[c];
        if (v) atomicAdd(&tile_counts[c], v);
    }
}

// K5 from the per-tile counts: ranges[t] = [start, start + count) with start = pairs in tiles before t; tiles
// without pairs keep (0, 0), as the reference's lists have it.  One 1024-thread workgroup.
// The counters are zeroed again as they are read, ready for the next view's histogram (no memset per view).
__device__ __forceinline__ void ranges_from_counts_body(uint32_t ntiles, uint32_t *__restrict__ tile_counts,
                                                        int32_t *__restrict__ ranges) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 1024u * 4u) {
        uint32_t c[4], sum = 0;
        const uint32_t t0 = base + threadIdx.x * 4u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = t0 + k < ntiles ? tile_counts[t0 + k] : 0u;
            if (c[k]) tile_counts[t0 + k] = 0u;
            sum += c[k];
        }
        const uint32_t inc = wave_inclusive_scan(sum);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, chunk = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) wbase += s_wave[w];
            chunk += s_wave[w];
        }
        uint32_t run = s_carry + wbase + inc - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (t0 + k < ntiles) {
                ranges[2 * (t0 + k)] = c[k] ? (int32_t)run : 0;
                ranges[2 * (t0 + k) + 1] = c[k] ? (int32_t)(run + c[k]) : 0;
            }
            run += c[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += chunk;
        __syncthreads();
    }
}

// one 256-thread workgroup per digit: table[d][b] <- elements with digit d in blocks before b; totals[d] <- elements
// with digit d.  (No atomics: thousands of same-address atomics serialise at ~12 ns each.)
__device__ __forceinline__ void radix_scan_body(int bid, uint32_t nblocks, uint32_t *__restrict__ table,
                                                uint32_t *__restrict__ totals) {
    const uint32_t d = bid;
    uint32_t *row = table + (size_t)d * nblocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += SCAN_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(v, &total);
        if (i < nblocks) row[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) totals[d] = carry;
}

// exclusive scan of one value per thread over the first 256 threads of a workgroup (any THREADS >= 256);
// every thread of the workgroup must call it
template <int THREADS>
__device__ __forceinline__ uint32_t scan256(uint32_t v, uint32_t *wave_sums /* [4] LDS */, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63 && wave < 4) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t s = wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <int NBITS, int THREADS, int ITEMS, int MODE>
__device__ __forceinline__ void radix2_scatter_body(uint32_t slot, uint32_t n, RadixIO io, int shift, uint32_t mask,
                                                    uint32_t nblocks, const uint32_t *__restrict__ table,
                                                    const uint32_t *__restrict__ totals) {
    constexpr int TILE = THREADS * ITEMS, WAVES = THREADS / 64, WAVE_TILE = 64 * ITEMS;
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "a workgroup has at least the 256 digit threads");
    __shared__ uint32_t s_cnt[WAVES][RADIX_BINS];   // per-wave digit counts -> exclusive wave bases
    __shared__ uint32_t s_lbase[RADIX_BINS];        // where digit d starts inside this block
    __shared__ uint32_t s_gbase[RADIX_BINS];        // global start of (d, block) minus s_lbase[d]
    __shared__ uint32_t s_ws[4];
    __shared__ uint2 s_el[TILE];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t bid = radix_logical_block(slot, nblocks);
    const uint32_t base = bid * (uint32_t)TILE;
    if (base >= n) return;     // whole workgroup (a grid sized by an upper bound of n); before any barrier
    const uint32_t wbase = base + (uint32_t)w * WAVE_TILE;
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (int c = t; c < WAVES * RADIX_BINS; c += THREADS) (&s_cnt[0][0])[c] = 0u;

    // wave-striped arrangement: the block's element order is (wave, item, lane)
    uint32_t key[ITEMS], rank[ITEMS];
    int32_t val[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = wbase + k * 64 + lane;
        key[k] = 0xFFFFFFFFu;
        val[k] = 0;
        if (i < n) {
            if (MODE == RADIX_SOA) {
                key[k] = io.keys_in[i];
                val[k] = io.vals_in ? io.vals_in[i] : (int32_t)i;
            } else {
                const uint2 p = io.pairs_in[i];
                key[k] = p.x;
                val[k] = (int32_t)p.y;
            }
        }
    }
    __syncthreads();

    // plain LDS accesses (a volatile pointer decays to a generic one: flat loads with a full wait).  LDS
    // operations of one wave execute in order; the fences keep the compiler from moving them across each other.
    uint32_t *cnt = &s_cnt[w][0];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = wbase + k * 64 + lane < n;
        const uint32_t d = (key[k] >> shift) & mask;
        // lanes of this wave holding the same digit: per bit, keep the lanes whose bit equals mine
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

    {   // thread t < 256 = digit t
        uint32_t run = 0;
        if (t < RADIX_BINS) {
#pragma unroll
            for (int ww = 0; ww < WAVES; ww++) {
                const uint32_t c = s_cnt[ww][t];
                s_cnt[ww][t] = run;
                run += c;
            }
        }
        uint32_t block_total, all;
        const uint32_t excl = scan256<THREADS>(t < RADIX_BINS ? run : 0u, s_ws, &block_total);
        const uint32_t dbase = scan256<THREADS>(t < RADIX_BINS ? totals[t] : 0u, s_ws, &all);   // keys with a smaller digit
        if (t < RADIX_BINS) {
            s_lbase[t] = excl;
            s_gbase[t] = dbase + table[(size_t)t * nblocks + bid] - excl;
        }
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        if (wbase + k * 64 + lane < n) {
            const uint32_t d = (key[k] >> shift) & mask;
            const uint32_t pos = s_lbase[d] + s_cnt[w][d] + rank[k];
            s_el[pos] = make_uint2(key[k], (uint32_t)val[k]);
        }
    }
    __syncthreads();

    const uint32_t count = min((uint32_t)TILE, n - base);
    for (uint32_t i = t; i < count; i += THREADS) {
        const uint2 e = s_el[i];
        const uint32_t pos = s_gbase[(e.x >> shift) & mask] + i;
        if (MODE == RADIX_PAIRS) {
            io.pairs_out[pos] = e;
        } else {
            if (io.keys_out) io.keys_out[pos] = e.x;
            io.vals_out[pos] = (int32_t)e.y;
        }
    }
}

}  // namespace
