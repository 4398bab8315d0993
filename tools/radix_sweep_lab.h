// LABORATORY COPY of csrc/radix_sort.h with a one-sweep (decoupled look-back) variant of the tile sort added -- not
// part of the library.  Result (tools/sort_lab.hip, MI355X, 122.5 M pairs, 15 key bits): table + scan + scatter 1.43 ms,
// one sweep per pass 1.38-1.41 ms: the histogram kernel halves (0.47 -> 0.22 ms, one LDS atomic per pair) and the
// second histogram read disappears, but the two sweeps take 1.00-1.03 ms against 0.83 ms for the two scatters --
// a status poll under a bandwidth-bound kernel takes microseconds while a chain receives a new block every
// ~0.25 us, so the walks stay tens of blocks deep (1 KiB of uncached status per step) whatever the chain length,
// the look-ahead or the overlap with the reorder; without look-back (wrong positions) the sweeps take 0.81 ms.
// Not adopted.
//
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
#include "../gaussian-splatting_amd/csrc/binning_kernels.h"

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

// One sweep per pass (SWEEP = true: the tile sort, up to two passes, for lists of tens of millions of pairs): no
// [digit][block] table, no scan kernel, one LDS atomic per pair in the histogram kernel and no second read of the
// pairs for the second pass's histogram.
//  * Blocks are taken in CHAINS: runs of consecutive blocks whose digit counts are known as a whole before the
//    sweep starts, so the elements in front of a chain are a table lookup and only the position inside the chain is
//    found at run time.  First pass: each of the 8 XCD-local ranges of blocks is cut into SWEEP_CHUNKS contiguous
//    chunks, one per histogram workgroup, which counts its pairs per tile id in LDS (K5's input) and leaves its
//    chunk's first-digit counts.  Second pass: sweep_plan_body groups consecutive first digits into at most 256
//    groups whose runs start on a block boundary of the intermediate array (the gaps, < 1 block each, are never
//    read: a group's last block knows how many pairs it holds); a group holds WHOLE first-digit runs, so its second-digit counts follow exactly from the tile counts.
//  * A scatter workgroup takes the next block of its XCD range from a ticket counter (of another range once its
//    own is used up), publishes its digit counts in a status word per (block, digit) and finds the elements of its
//    chain in front of it by decoupled look-back over its predecessors' status words (flag and count in one
//    32-bit word, relaxed agent-scope atomics).  A predecessor took its ticket earlier, so it is running, and it
//    publishes its aggregate before it looks back itself: every walk ends at the chain's first block, which never
//    waits.  Chains are short (n / 256 pairs) because a status poll under a bandwidth-bound kernel takes
//    microseconds: with 8 long chains the walks grew to hundreds of blocks (measured: 1.06 ms for the two sweeps at
//    122 M pairs against 0.81 ms without any look-back).  Every poll loop is bounded (SWEEP_SPIN_LIMIT): a broken
//    chain sets state[SWEEP_ERROR] and the kernel finishes with wrong positions (guarded against the array end)
//    instead of hanging the GPU.
constexpr uint32_t SWEEP_AGG = 1u << 30, SWEEP_INC = 2u << 30, SWEEP_VAL = (1u << 30) - 1u;
constexpr int SWEEP_CHUNKS = 32;                                      // chains per XCD range in the first pass
constexpr int SWEEP_CHAINS = 8 * SWEEP_CHUNKS;                        // = histogram workgroups; max groups of pass 2
constexpr uint32_t SWEEP_PAD_BLOCKS = SWEEP_CHAINS;                   // blocks the group alignment can add
constexpr uint32_t SWEEP_TICKET = 0;                                  // [pass][8]
constexpr uint32_t SWEEP_ERROR = 16;                                  // one word
constexpr uint32_t SWEEP_XSTART = 32;                                 // [pass][8] first block of the XCD range
constexpr uint32_t SWEEP_XCOUNT = 48;                                 // [pass][8] blocks of the XCD range
constexpr uint32_t SWEEP_CHUNK_LEN = 64;                              // [8] blocks per chunk of the range (pass 0)
constexpr uint32_t SWEEP_DBASE = 128;                                 // [pass][256] start of each digit's run
constexpr uint32_t SWEEP_GSIZE = 640;                                 // [256] pairs of each group of pass 1
constexpr uint32_t SWEEP_PRE = 1024;                                  // [pass][chain][256] elements before the chain
constexpr uint32_t SWEEP_BLKINFO = SWEEP_PRE + 2 * SWEEP_CHAINS * RADIX_BINS;   // [nblocks + pad] pass 1: group << 20 | index
constexpr uint32_t SWEEP_SPIN_LIMIT = 1u << 20;                       // x s_sleep(1): tens of milliseconds
#ifndef SWEEP_LOOK
#define SWEEP_LOOK 4                                                  // predecessors inspected per look-back round
#endif

__host__ __device__ inline size_t sweep_status_offset(uint32_t nblocks) {       // [pass][nblocks + pad][256]
    return ((size_t)SWEEP_BLKINFO + nblocks + SWEEP_PAD_BLOCKS + 255u) & ~(size_t)255u;
}
// words of state a sort over `nblocks` blocks needs (zeroed before the histogram kernel)
__host__ __device__ inline size_t sweep_state_words(uint32_t nblocks) {
    return sweep_status_offset(nblocks) + (size_t)2 * (nblocks + SWEEP_PAD_BLOCKS) * RADIX_BINS;
}
// pairs of the intermediate array (what the first sweep writes and the second reads)
__host__ __device__ inline size_t sweep_mid_capacity(uint32_t n, uint32_t tile) {
    return (size_t)n + (size_t)SWEEP_PAD_BLOCKS * tile;
}

struct SweepDigits {         // bits1 low bits of the tile id in the first pass, bits2 (0: one pass) in the second
    int bits1, bits2;
};

// range r of the 8: its first logical block and its number of blocks (the numbering of radix_logical_block)
__device__ __forceinline__ void sweep_range(uint32_t r, uint32_t nblocks, uint32_t &start, uint32_t &count) {
    const uint32_t per = nblocks >> 3, rem = nblocks & 7u;
    start = r * per + min(r, rem);
    count = per + (r < rem ? 1u : 0u);
}

// launch slot -> logical block: consecutive logical blocks share an XCD (see the header comment)
__device__ __forceinline__ uint32_t radix_logical_block(uint32_t slot, uint32_t nblocks) {
#ifdef GS_RADIX_NO_XCD_REMAP
    return slot;
#endif
    const uint32_t per = nblocks >> 3, rem = nblocks & 7u, x = slot & 7u, j = slot >> 3;
    return x * per + min(x, rem) + j;
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

// The histogram kernel of the one-sweep sort, SWEEP_CHAINS workgroups of 1024 threads: workgroup w counts the pairs
// of chunk w / 8 of XCD range w & 7 per tile id in LDS (one LDS atomic per pair), adds them to tile_counts[] and
// leaves the chunk's first-digit counts (a digit is a bit field of the tile id) in its own row of the table.
template <int THREADS, int ITEMS>
__device__ __forceinline__ void sweep_hist_tiles_body(uint32_t n, const uint2 *__restrict__ pairs, uint32_t nblocks,
                                                      SweepDigits dg, uint32_t *__restrict__ state, uint32_t ntiles,
                                                      uint32_t *__restrict__ tile_counts, uint32_t *s_tiles) {
    constexpr int TILE = THREADS * ITEMS;        // the block size of the sort's scatter kernels
    __shared__ uint32_t hd[RADIX_BINS];
    for (uint32_t c = threadIdx.x; c < ntiles; c += THREADS) s_tiles[c] = 0u;
    for (int c = threadIdx.x; c < RADIX_BINS; c += THREADS) hd[c] = 0u;
    __syncthreads();
    const uint32_t x = blockIdx.x & 7u, chunk = blockIdx.x >> 3;
    uint32_t xstart, xcount;
    sweep_range(x, nblocks, xstart, xcount);
    const uint32_t len = (xcount + SWEEP_CHUNKS - 1u) / SWEEP_CHUNKS;
    const uint32_t b0 = min(chunk * len, xcount), b1 = min(b0 + len, xcount);
    for (uint32_t bid = xstart + b0; bid < xstart + b1; bid++) {
        const uint32_t base = bid * (uint32_t)TILE;
        const uint4 *p4 = reinterpret_cast<const uint4 *>(pairs + base);     // two pairs per 16-byte load
#pragma unroll
        for (int k = 0; k < ITEMS / 2; k++) {
            const uint32_t e = (k * THREADS + threadIdx.x) * 2u, i = base + e;
            if (i + 1 < n) {
                const uint4 v = p4[k * THREADS + threadIdx.x];
                if (v.x < ntiles) atomicAdd(&s_tiles[v.x], 1u);
                if (v.z < ntiles) atomicAdd(&s_tiles[v.z], 1u);
            } else if (i < n) {
                const uint32_t key = pairs[i].x;
                if (key < ntiles) atomicAdd(&s_tiles[key], 1u);
            }
        }
    }
    __syncthreads();
    const uint32_t mask1 = (1u << dg.bits1) - 1u;
    for (uint32_t c = threadIdx.x; c < ntiles; c += THREADS) {
        const uint32_t v = s_tiles[c];
        if (v) {
            atomicAdd(&tile_counts[c], v);
            atomicAdd(&hd[c & mask1], v);
        }
    }
    __syncthreads();
    uint32_t *row = state + SWEEP_PRE + (size_t)(x * SWEEP_CHUNKS + chunk) * RADIX_BINS;      // pass 0, my chain
    for (int d = threadIdx.x; d < RADIX_BINS; d += THREADS) row[d] = hd[d];
}

// rows[chain][256] (counts per chain and digit) -> elements of that digit in the chains before, in place.  1024
// threads: thread (digit, quarter) sums its quarter of the chains, the quarters are joined through LDS, then it
// rewrites its quarter; 16 independent loads at a time (the workgroup is alone on its CU: latency is all there is).
__device__ __forceinline__ void sweep_prefix_rows(uint32_t *__restrict__ rows, uint32_t *s_part /* [4][256] */) {
    constexpr int Q = SWEEP_CHAINS / 4;
    static_assert(Q % 16 == 0, "quarters in batches of 16 chains");
    const uint32_t d = threadIdx.x & 255u, part = threadIdx.x >> 8;
    uint32_t *col = rows + (size_t)part * Q * RADIX_BINS + d;
    uint32_t sum = 0;
    for (int c0 = 0; c0 < Q; c0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = col[(size_t)(c0 + j) * RADIX_BINS];
#pragma unroll
        for (int j = 0; j < 16; j++) sum += v[j];
    }
    s_part[part * 256 + d] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t q = 0; q < part; q++) run += s_part[q * 256 + d];
    for (int c0 = 0; c0 < Q; c0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = col[(size_t)(c0 + j) * RADIX_BINS];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            col[(size_t)(c0 + j) * RADIX_BINS] = run;
            run += v[j];
        }
    }
    __syncthreads();
}

// One 1024-thread workgroup between the histogram kernel and the sweeps (see the header comment); with one pass
// (bits2 == 0) the first digit's runs are packed and there are no groups.  Everything is a scan or an LDS atomic:
// digit d belongs to group floor(start of its run / target), target = n / 256 + 1 (at least 8 blocks), so a group is a
// set of consecutive digits, there are at most 256 group ids (some may stay empty) and the id is the chain index.
__device__ __forceinline__ void sweep_plan_body(uint32_t n, uint32_t nblocks, uint32_t tile, SweepDigits dg,
                                                uint32_t ntiles, const uint32_t *__restrict__ tile_counts,
                                                uint32_t *__restrict__ state) {
    __shared__ uint32_t T1[RADIX_BINS], T2[RADIX_BINS], grp[RADIX_BINS], s_part[4 * 256], s_ws[4];
    __shared__ uint32_t gfirst[SWEEP_CHAINS], gsize[SWEEP_CHAINS], gblock[SWEEP_CHAINS], xfirst[8], xcount[8];
    const uint32_t mask1 = (1u << dg.bits1) - 1u, t = threadIdx.x;
    const bool two = dg.bits2 > 0;
    if (t < RADIX_BINS) { T1[t] = 0u; T2[t] = 0u; gfirst[t] = 0xFFFFFFFFu; gsize[t] = 0u; }
    if (t < 8u) { xfirst[t] = 0xFFFFFFFFu; xcount[t] = 0u; }
    __syncthreads();
    for (uint32_t k = t; k < ntiles; k += 1024) {
        const uint32_t v = tile_counts[k];
        if (v) {
            atomicAdd(&T1[k & mask1], v);
            atomicAdd(&T2[k >> dg.bits1], v);
        }
    }
    __syncthreads();
    uint32_t total;
    const uint32_t mine1 = t < RADIX_BINS ? T1[t] : 0u, mine2 = t < RADIX_BINS ? T2[t] : 0u;
    const uint32_t p1 = scan256<1024>(mine1, s_ws, &total);      // start of digit t's run in a packed array
    const uint32_t p2 = scan256<1024>(mine2, s_ws, &total);
    if (t < RADIX_BINS) state[SWEEP_DBASE + RADIX_BINS + t] = p2;   // the last pass's output is packed
    if (t < 8u) {     // pass 0: XCD ranges by blocks of the input, cut into SWEEP_CHUNKS chunks
        uint32_t start, count;
        sweep_range(t, nblocks, start, count);
        state[SWEEP_XSTART + t] = start;
        state[SWEEP_XCOUNT + t] = count;
        state[SWEEP_CHUNK_LEN + t] = (count + SWEEP_CHUNKS - 1u) / SWEEP_CHUNKS;
    }
    if (!two) {
        if (t < RADIX_BINS) state[SWEEP_DBASE + t] = p1;
        sweep_prefix_rows(state + SWEEP_PRE, s_part);
        return;
    }
    const uint32_t target = max(n / (uint32_t)SWEEP_CHAINS + 1u, 8u * tile);
    const uint32_t g = p1 / target;       // < 256 (p1 < n)
    if (t < RADIX_BINS) {
        grp[t] = g;
        if (mine1) {
            atomicMin(&gfirst[g], p1);
            atomicAdd(&gsize[g], mine1);
        }
    }
    __syncthreads();
    // blocks per group -> first block of each group (groups are laid out one after the other, each from a block boundary)
    const uint32_t nb = t < RADIX_BINS ? (gsize[t] + tile - 1u) / tile : 0u;
    uint32_t total_blocks;
    const uint32_t b0 = scan256<1024>(nb, s_ws, &total_blocks);
    if (t < RADIX_BINS) gblock[t] = b0;
    __syncthreads();
    if (t < RADIX_BINS) {
        if (mine1) state[SWEEP_DBASE + t] = gblock[g] * tile + (p1 - gfirst[g]);
        else state[SWEEP_DBASE + t] = 0u;
        // thread t as group t: its XCD range (an eighth of the blocks each, whole groups), its size, its blocks
        state[SWEEP_GSIZE + t] = gsize[t];
        if (nb) {
            const uint32_t x = (uint32_t)(((unsigned long long)b0 * 8ull) / total_blocks);
            atomicMin(&xfirst[x], b0);
            atomicAdd(&xcount[x], nb);
            for (uint32_t k = 0; k < nb; k++) state[SWEEP_BLKINFO + b0 + k] = t << 20 | k;
        }
    }
    __syncthreads();
    if (t < 8u) {
        state[SWEEP_XSTART + 8 + t] = xcount[t] ? xfirst[t] : 0u;
        state[SWEEP_XCOUNT + 8 + t] = xcount[t];
    }
    // pass 0: counts per chain -> elements in the chains before
    sweep_prefix_rows(state + SWEEP_PRE, s_part);
    // pass 1: second-digit counts per group straight from the tile counts (the rows were zeroed with the state)
    uint32_t *rows2 = state + SWEEP_PRE + (size_t)SWEEP_CHAINS * RADIX_BINS;
    for (uint32_t k = t; k < ntiles; k += 1024) {
        const uint32_t v = tile_counts[k];
        if (v) atomicAdd(&rows2[(size_t)grp[k & mask1] * RADIX_BINS + (k >> dg.bits1)], v);
    }
    __threadfence();
    __syncthreads();
    sweep_prefix_rows(rows2, s_part);
}

// K5 from the per-tile counts: ranges[t] = [start, start + count) with start = pairs in tiles before t; tiles
// without pairs keep (0, 0), as the reference's lists have it.  One 1024-thread workgroup.
__device__ __forceinline__ void ranges_from_counts_body(uint32_t ntiles, const uint32_t *__restrict__ tile_counts,
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

template <int NBITS, int THREADS, int ITEMS, int MODE, bool SWEEP>
__device__ __forceinline__ void radix2_scatter_impl(uint32_t slot, uint32_t n, RadixIO io, int shift, uint32_t mask,
                                                    uint32_t nblocks, const uint32_t *__restrict__ table,
                                                    const uint32_t *__restrict__ totals, int pass,
                                                    uint32_t *__restrict__ state, uint32_t out_cap) {
    constexpr int TILE = THREADS * ITEMS, WAVES = THREADS / 64, WAVE_TILE = 64 * ITEMS;
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "a workgroup has at least the 256 digit threads");
    __shared__ uint32_t s_cnt[WAVES][RADIX_BINS];   // per-wave digit counts -> exclusive wave bases
    __shared__ uint32_t s_lbase[RADIX_BINS];        // where digit d starts inside this block
    __shared__ uint32_t s_gbase[RADIX_BINS];        // global start of (d, block) minus s_lbase[d]
    __shared__ uint32_t s_ws[4];
    __shared__ uint2 s_el[TILE];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t bid, chain = 0, index = 0;      // SWEEP: my chain and my place in it
#ifdef SWEEP_STATIC0
    if (SWEEP && pass == 0) {  // experiment: no ticket in the first pass (in-order dispatch assumed)
        const uint32_t x = slot & 7u, tk = slot >> 3;
        uint32_t xs, xc;
        sweep_range(x, nblocks, xs, xc);
        const uint32_t len = (xc + SWEEP_CHUNKS - 1u) / SWEEP_CHUNKS, c = tk / len;
        bid = xs + tk;
        chain = x * SWEEP_CHUNKS + c;
        index = tk - c * len;
    } else
#endif
    if (SWEEP) {               // the next block of this slot's XCD range (of another range once that one is used up)
        __shared__ uint32_t s_ticket[4];
        if (t == 0) {
            uint32_t x = slot & 7u, tk = 0;
            int k = 0;
            for (; k < 8; k++, x = (x + 1u) & 7u) {
                if (state[SWEEP_XCOUNT + pass * 8 + x] == 0u) continue;
                tk = atomicAdd(&state[SWEEP_TICKET + pass * 8 + x], 1u);
                if (tk < state[SWEEP_XCOUNT + pass * 8 + x]) break;
            }
            if (k == 8) {
                s_ticket[0] = 0xFFFFFFFFu;
            } else {
                const uint32_t b = state[SWEEP_XSTART + pass * 8 + x] + tk;
                s_ticket[0] = b;
                if (pass == 0) {
                    const uint32_t len = state[SWEEP_CHUNK_LEN + x], c = tk / len;
                    s_ticket[1] = x * SWEEP_CHUNKS + c;
                    s_ticket[2] = tk - c * len;
                    s_ticket[3] = n;
                } else {
                    const uint32_t info = state[SWEEP_BLKINFO + b], g = info >> 20, kk = info & 0xFFFFFu;
                    s_ticket[1] = g;
                    s_ticket[2] = kk;
                    // pairs of the block: the group's last block is partly gap
                    s_ticket[3] = b * (uint32_t)TILE + min((uint32_t)TILE, state[SWEEP_GSIZE + g] - kk * (uint32_t)TILE);
                }
            }
        }
        __syncthreads();
        bid = s_ticket[0];
        chain = s_ticket[1];
        index = s_ticket[2];
        if (bid == 0xFFFFFFFFu) return;     // more workgroups than blocks
        n = s_ticket[3];                    // elements below this index are the block's pairs
    } else {
        bid = radix_logical_block(slot, nblocks);
    }
    const uint32_t base = bid * (uint32_t)TILE;
    if (!SWEEP && base >= n) return;     // whole workgroup (a grid sized by an upper bound of n); before any barrier
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

    // thread t < 256 = digit t
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
    // SWEEP: the look-back is split around the reorder below -- publish my counts and ask for the nearest
    // predecessors' status words first, consume the answers afterwards (a status poll takes microseconds under a
    // bandwidth-bound kernel; the reorder hides the first round)
    uint32_t *st = nullptr;
    uint32_t lb[SWEEP_LOOK];
    if (!SWEEP) {
        const uint32_t dbase = scan256<THREADS>(t < RADIX_BINS ? totals[t] : 0u, s_ws, &all);   // keys with a smaller digit
        if (t < RADIX_BINS) {
            s_lbase[t] = excl;
            s_gbase[t] = dbase + table[(size_t)t * nblocks + bid] - excl;
        }
    } else if (t < RADIX_BINS) {
        s_lbase[t] = excl;
        st = state + sweep_status_offset(nblocks) + ((size_t)pass * (nblocks + SWEEP_PAD_BLOCKS) + bid) * RADIX_BINS + t;
        __hip_atomic_store(st, (index == 0 ? SWEEP_INC : SWEEP_AGG) | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < SWEEP_LOOK; j++)
            lb[j] = (uint32_t)j < index ? __hip_atomic_load(st - (j + 1) * RADIX_BINS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : SWEEP_INC;
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

    if (SWEEP && t < RADIX_BINS) {
        // keys with a smaller digit and keys of this digit in the chains before mine: from the plan
        const uint32_t dbase = state[SWEEP_DBASE + pass * RADIX_BINS + t];
        const uint32_t before = state[SWEEP_PRE + ((size_t)pass * SWEEP_CHAINS + chain) * RADIX_BINS + t];
        // keys of this digit in the blocks of my chain before mine: decoupled look-back, SWEEP_LOOK predecessors per
        // round (their status words are loaded together, then consumed nearest first)
        uint32_t in_front = 0;
        if (index != 0) {
            const uint32_t *q = st;
            uint32_t left = index;
            bool open = true;
#ifdef SWEEP_NO_LOOKBACK
            open = false;      // experiment: the sweep without its look-back (wrong positions)
#endif
            while (open) {
#pragma unroll
                for (int j = 0; j < SWEEP_LOOK; j++) {
                    if (!open) break;
                    uint32_t spins = 0;
                    while ((lb[j] >> 30) == 0u) {
                        __builtin_amdgcn_s_sleep(1);
                        lb[j] = __hip_atomic_load(q - (j + 1) * RADIX_BINS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (++spins > SWEEP_SPIN_LIMIT) {      // never in a correct run: give up, do not hang
                            atomicOr(&state[SWEEP_ERROR], 1u);
                            lb[j] = SWEEP_INC;
                        }
                    }
                    in_front += lb[j] & SWEEP_VAL;
                    if ((lb[j] >> 30) == 2u) open = false;
                }
                if (!open) break;
                q -= SWEEP_LOOK * RADIX_BINS;
                left -= SWEEP_LOOK;        // (open: all SWEEP_LOOK were real predecessors, so left > SWEEP_LOOK)
#pragma unroll
                for (int j = 0; j < SWEEP_LOOK; j++)
                    lb[j] = (uint32_t)j < left ? __hip_atomic_load(q - (j + 1) * RADIX_BINS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : SWEEP_INC;
            }
            __hip_atomic_store(st, SWEEP_INC | (in_front + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gbase[t] = dbase + before + in_front - excl;
    }
    __syncthreads();

    const uint32_t count = min((uint32_t)TILE, n - base);
    for (uint32_t i = t; i < count; i += THREADS) {
        const uint2 e = s_el[i];
        const uint32_t pos = s_gbase[(e.x >> shift) & mask] + i;
        if (SWEEP && pos >= out_cap) continue;      // only after a broken look-back chain (state[SWEEP_ERROR])
        if (MODE == RADIX_PAIRS) {
            io.pairs_out[pos] = e;
        } else {
            if (io.keys_out) io.keys_out[pos] = e.x;
            io.vals_out[pos] = (int32_t)e.y;
        }
    }
}

template <int NBITS, int THREADS, int ITEMS, int MODE>
__device__ __forceinline__ void radix2_scatter_body(uint32_t slot, uint32_t n, RadixIO io, int shift, uint32_t mask,
                                                    uint32_t nblocks, const uint32_t *__restrict__ table,
                                                    const uint32_t *__restrict__ totals) {
    radix2_scatter_impl<NBITS, THREADS, ITEMS, MODE, false>(slot, n, io, shift, mask, nblocks, table, totals, 0, nullptr, 0u);
}

template <int NBITS, int THREADS, int ITEMS, int MODE>
// n: elements of the input array that may be read (pass 0: the pairs; pass 1: the intermediate array's capacity,
// every element of a block there is a pair or a sentinel); out_cap: elements of the output array
__device__ __forceinline__ void radix2_sweep_body(uint32_t slot, uint32_t n, RadixIO io, int shift, uint32_t mask,
                                                  uint32_t nblocks, int pass, uint32_t *__restrict__ state,
                                                  uint32_t out_cap) {
    radix2_scatter_impl<NBITS, THREADS, ITEMS, MODE, true>(slot, n, io, shift, mask, nblocks, nullptr, nullptr, pass, state,
                                                           out_cap);
}

}  // namespace
