// Laboratory for the tile sort (not part of the library): times the radix pass variants of csrc/radix_sort.h on
// synthetic (tile id, gaussian) pairs and checks them against std::stable_sort.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/sort_lab.hip -o tools/sort_lab && tools/sort_lab [D] [tiles]
//   tools/sort_lab check        a sweep of awkward sizes through both variants, checked only
#include "radix_sweep_lab.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(SCAN_THREADS) void scan_k(uint32_t nblocks, uint32_t *table, uint32_t *totals) {
    radix_scan_body(blockIdx.x, nblocks, table, totals);
}
template <int THREADS, int ITEMS, bool PAIRS>
__global__ __launch_bounds__(THREADS) void hist_k(uint32_t n, const uint32_t *keys, const uint2 *pairs, int shift, uint32_t mask, uint32_t nblocks, uint32_t *table) {
    radix2_hist_body<THREADS, ITEMS, PAIRS>(blockIdx.x, n, keys, pairs, shift, mask, nblocks, table);
}
template <int ITEMS_PER_1024>
__global__ __launch_bounds__(1024) void tile_hist_k(uint32_t n, const uint2 *pairs, uint32_t mask, uint32_t nblocks, uint32_t *table, uint32_t ntiles, uint32_t *tile_counts) {
    extern __shared__ uint32_t s_dyn[];
    radix2_hist_tiles_body<1024, ITEMS_PER_1024>(n, pairs, mask, nblocks, table, ntiles, tile_counts, s_dyn);
}
template <int NBITS, int THREADS, int ITEMS, int MODE>
__global__ __launch_bounds__(THREADS) void scatter_k(uint32_t n, RadixIO io, int shift, uint32_t mask, uint32_t nblocks, const uint32_t *table, const uint32_t *totals) {
    radix2_scatter_body<NBITS, THREADS, ITEMS, MODE>(blockIdx.x, n, io, shift, mask, nblocks, table, totals);
}
// ---- one sweep per pass
template <int ITEMS_PER_1024>
__global__ __launch_bounds__(1024) void sweep_hist_k(uint32_t n, const uint2 *pairs, uint32_t nblocks, SweepDigits dg, uint32_t *state, uint32_t ntiles, uint32_t *tile_counts) {
    extern __shared__ uint32_t s_dyn[];
    sweep_hist_tiles_body<1024, ITEMS_PER_1024>(n, pairs, nblocks, dg, state, ntiles, tile_counts, s_dyn);
}
__global__ __launch_bounds__(1024) void plan_k(uint32_t n, uint32_t nblocks, uint32_t tile, SweepDigits dg, uint32_t ntiles, const uint32_t *tile_counts, uint32_t *state) {
    sweep_plan_body(n, nblocks, tile, dg, ntiles, tile_counts, state);
}
template <int NBITS, int THREADS, int ITEMS, int MODE>
__global__ __launch_bounds__(THREADS) void sweep_k(uint32_t n, RadixIO io, int shift, uint32_t mask, uint32_t nblocks, int pass, uint32_t *state, uint32_t out_cap) {
    radix2_sweep_body<NBITS, THREADS, ITEMS, MODE>(blockIdx.x, n, io, shift, mask, nblocks, pass, state, out_cap);
}

struct Bufs {
    uint2 *p[2]; uint32_t *table; int32_t *out_vals; uint32_t *tile_counts; uint32_t *state; size_t state_words;
};

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    void start() { CK(hipEventRecord(a, 0)); }
    float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

static int bits_of(uint32_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

template <typename K> static void allow_lds(K kern) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
}

// the library's sort as it is: tile histogram + table + scan + scatter per pass; t = {hist, scan, scatter}
template <int THREADS, int ITEMS>
void sort_table(Bufs &B, uint32_t n, uint32_t tiles, int bits, float t[3], bool timeit) {
    constexpr int TILE = THREADS * ITEMS;
    const uint32_t nblocks = (n + TILE - 1) / TILE;
    uint32_t *totals = B.table, *table = B.table + RADIX_BINS;
    const int passes = (bits + 7) / 8, per = (bits + passes - 1) / passes;
    Timer T;
    CK(hipMemsetAsync(B.tile_counts, 0, (size_t)tiles * 4, 0));
    for (int p = 0; p < passes; p++) {
        const int shift = p * per, nb = std::min(per, bits - shift);
        const uint32_t mask = (1u << nb) - 1u;
        const bool last = p == passes - 1;
        RadixIO io{};
        io.pairs_in = B.p[p & 1]; io.pairs_out = B.p[(p & 1) ^ 1]; io.vals_out = B.out_vals;
        if (timeit) T.start();
        if (p == 0) {
            allow_lds(tile_hist_k<TILE / 1024>);
            hipLaunchKernelGGL((tile_hist_k<TILE / 1024>), dim3(std::min(nblocks, 256u)), dim3(1024), (size_t)tiles * 4, 0, n, io.pairs_in, mask, nblocks, table, tiles, B.tile_counts);
        } else {
            hipLaunchKernelGGL((hist_k<THREADS, ITEMS, true>), dim3(nblocks), dim3(THREADS), 0, 0, n, (const uint32_t *)nullptr, io.pairs_in, shift, mask, nblocks, table);
        }
        if (timeit) t[0] += T.stop();
        if (timeit) T.start();
        hipLaunchKernelGGL(scan_k, dim3(RADIX_BINS), dim3(SCAN_THREADS), 0, 0, nblocks, table, totals);
        if (timeit) t[1] += T.stop();
        if (timeit) T.start();
#define LAUNCH(NB, MODE) hipLaunchKernelGGL((scatter_k<NB, THREADS, ITEMS, MODE>), dim3(nblocks), dim3(THREADS), 0, 0, n, io, shift, mask, nblocks, (const uint32_t *)table, (const uint32_t *)totals)
#define BY_MODE(NB) do { if (last) LAUNCH(NB, RADIX_PAIRS_TO_SOA); else LAUNCH(NB, RADIX_PAIRS); } while (0)
        switch (nb) { case 1: BY_MODE(1); break; case 4: BY_MODE(4); break; case 5: BY_MODE(5); break; case 6: BY_MODE(6); break; case 7: BY_MODE(7); break; case 8: BY_MODE(8); break; default: printf("nbits %d not instantiated\n", nb); exit(1); }
#undef LAUNCH
        if (timeit) t[2] += T.stop();
    }
    CK(hipGetLastError());
}

// one sweep per pass: state memset + tile histogram + plan + one kernel per pass; t = {hist, memset + plan, sweeps}
template <int THREADS, int ITEMS>
void sort_sweep(Bufs &B, uint32_t n, uint32_t tiles, int bits, float t[3], bool timeit) {
    constexpr int TILE = THREADS * ITEMS;
    const uint32_t nblocks = (n + TILE - 1) / TILE;
    const int passes = (bits + 7) / 8, per = (bits + passes - 1) / passes;
    if (passes > 2) { printf("sweep: more than two passes\n"); exit(1); }
    SweepDigits dg{per, passes == 2 ? bits - per : 0};
    if (sweep_state_words(nblocks) > B.state_words) { printf("state too small\n"); exit(1); }
    const uint32_t cap = (uint32_t)sweep_mid_capacity(n, TILE);      // the intermediate array
    Timer T;
    if (timeit) T.start();
    CK(hipMemsetAsync(B.tile_counts, 0, (size_t)tiles * 4, 0));
    CK(hipMemsetAsync(B.state, 0, sweep_state_words(nblocks) * 4, 0));
    if (timeit) t[1] += T.stop();
    if (timeit) T.start();
    allow_lds(sweep_hist_k<TILE / 1024>);
    hipLaunchKernelGGL((sweep_hist_k<TILE / 1024>), dim3(SWEEP_CHAINS), dim3(1024), (size_t)tiles * 4, 0, n, B.p[0], nblocks, dg, B.state, tiles, B.tile_counts);
    if (timeit) t[0] += T.stop();
    if (timeit) T.start();
    hipLaunchKernelGGL(plan_k, dim3(1), dim3(1024), 0, 0, n, nblocks, (uint32_t)TILE, dg, tiles, (const uint32_t *)B.tile_counts, B.state);
    if (timeit) t[1] += T.stop();
    for (int p = 0; p < passes; p++) {
        const int shift = p * per, nb = p == 0 ? dg.bits1 : dg.bits2;
        const uint32_t mask = (1u << nb) - 1u;
        const bool last = p == passes - 1;
        RadixIO io{};
        io.pairs_in = B.p[p & 1]; io.pairs_out = B.p[(p & 1) ^ 1]; io.vals_out = B.out_vals;
        const uint32_t n_in = p == 0 ? n : cap, out_cap = last ? n : cap;
        if (timeit) T.start();
#define LAUNCH(NB, MODE) hipLaunchKernelGGL((sweep_k<NB, THREADS, ITEMS, MODE>), dim3(p == 0 ? nblocks : nblocks + SWEEP_PAD_BLOCKS), dim3(THREADS), 0, 0, n_in, io, shift, mask, nblocks, p, B.state, out_cap)
        switch (nb) { case 1: BY_MODE(1); break; case 4: BY_MODE(4); break; case 5: BY_MODE(5); break; case 6: BY_MODE(6); break; case 7: BY_MODE(7); break; case 8: BY_MODE(8); break; default: printf("nbits %d not instantiated\n", nb); exit(1); }
        if (timeit) t[2] += T.stop();
    }
    CK(hipGetLastError());
}

struct Case {
    uint32_t n, tiles;
    std::vector<uint2> hp;
    std::vector<int32_t> want;
    std::vector<uint32_t> want_counts;
};

static void make_case(Case &c, bool with_answer, int skew) {
    c.hp.resize(c.n);
    uint64_t s = 88172645463325252ull + c.n;
    for (uint32_t i = 0; i < c.n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t k = (uint32_t)(s % c.tiles);
        if (skew == 1) k = (uint32_t)((s >> 20) % 3 == 0 ? 0 : k);            // a third of the pairs in tile 0
        if (skew == 2) k = (uint32_t)((uint64_t)i * c.tiles / c.n);            // already sorted
        c.hp[i] = make_uint2(k, i);
    }
    if (!with_answer) return;
    c.want.resize(c.n);
    std::iota(c.want.begin(), c.want.end(), 0);
    std::stable_sort(c.want.begin(), c.want.end(), [&](int32_t a, int32_t b) { return c.hp[a].x < c.hp[b].x; });
    c.want_counts.assign(c.tiles, 0);
    for (uint32_t i = 0; i < c.n; i++) c.want_counts[c.hp[i].x]++;
}

static bool verify(const Case &c, Bufs &B, const char *name) {
    std::vector<int32_t> got(c.n);
    std::vector<uint32_t> cnt(c.tiles);
    uint32_t err = 0;
    CK(hipMemcpy(got.data(), B.out_vals, (size_t)c.n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cnt.data(), B.tile_counts, (size_t)c.tiles * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&err, B.state + SWEEP_ERROR, 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < c.n; i++)
        if (got[i] != c.want[i]) { printf("  %s n=%u tiles=%u: MISMATCH at %u: got %d want %d\n", name, c.n, c.tiles, i, got[i], c.want[i]); return false; }
    for (uint32_t i = 0; i < c.tiles; i++)
        if (cnt[i] != c.want_counts[i]) { printf("  %s n=%u: tile count %u: got %u want %u\n", name, c.n, i, cnt[i], c.want_counts[i]); return false; }
    if (err) { printf("  %s n=%u: look-back gave up (error word %u)\n", name, c.n, err); return false; }
    return true;
}

int main(int argc, char **argv) {
    const bool check_only = argc > 1 && !strcmp(argv[1], "check");
    const uint32_t n_max = check_only ? 9000000u : (argc > 1 ? (uint32_t)atoll(argv[1]) : 122469486u);
    const uint32_t tiles_arg = argc > 2 ? (uint32_t)atoi(argv[2]) : 32400u;
    Bufs B{};
    const size_t n8 = (sweep_mid_capacity(n_max, 8192) * 8 + 255) & ~(size_t)255;
    for (int i = 0; i < 2; i++) CK(hipMalloc(&B.p[i], n8));
    CK(hipMalloc(&B.out_vals, n8 / 2));
    CK(hipMalloc(&B.table, ((size_t)RADIX_BINS * (n_max / 4096 + 2) + RADIX_BINS) * 4));
    CK(hipMalloc(&B.tile_counts, 32768 * 4));
    B.state_words = sweep_state_words(n_max / 4096 + 2);
    CK(hipMalloc(&B.state, B.state_words * 4));
    CK(hipMemset(B.state, 0, B.state_words * 4));

    if (check_only) {
        const uint32_t sizes[] = {1, 2, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 32768, 32769, 65537, 100000, 1000003, 5604468, 8999999};
        const uint32_t tile_counts[] = {1, 2, 17, 256, 257, 920, 8160, 32400, 32768};
        int bad = 0, runs = 0;
        for (uint32_t n : sizes)
            for (uint32_t tiles : tile_counts) {
                if (n > 200000 && tiles != 8160 && tiles != 32400 && tiles != 2) continue;
                for (int skew = 0; skew < 3; skew++) {
                    if (skew && n < 4096) continue;
                    Case c{n, tiles};
                    make_case(c, true, skew);
                    const int bits = std::max(1, bits_of(tiles - 1));
                    float t[3];
                    auto one = [&](const char *name, auto fn) {
                        CK(hipMemcpy(B.p[0], c.hp.data(), (size_t)n * 8, hipMemcpyHostToDevice));
                        CK(hipMemset(B.out_vals, 0xff, (size_t)n * 4));
                        fn(t);
                        CK(hipDeviceSynchronize());
                        runs++;
                        if (!verify(c, B, name)) bad++;
                    };
                    one("table 256x16", [&](float *tt) { sort_table<256, 16>(B, n, tiles, bits, tt, false); });
                    one("sweep 256x16", [&](float *tt) { sort_sweep<256, 16>(B, n, tiles, bits, tt, false); });
                    one("sweep 512x16", [&](float *tt) { sort_sweep<512, 16>(B, n, tiles, bits, tt, false); });
                    if (bad > 5) { printf("too many failures\n"); return 1; }
                }
            }
        printf("check: %d sorts, %d bad\n", runs, bad);
        return bad ? 1 : 0;
    }

    Case c{n_max, tiles_arg};
    const bool check = n_max <= 8000000u;
    make_case(c, check, 0);
    const int bits = std::max(1, bits_of(c.tiles - 1));
    printf("n=%u tiles=%u bits=%d  algorithmic bytes 2*n*20 = %.1f MB\n", c.n, c.tiles, bits, 40.0 * c.n / 1e6);
    auto run = [&](const char *name, auto fn) {
        CK(hipMemcpy(B.p[0], c.hp.data(), (size_t)c.n * 8, hipMemcpyHostToDevice));
        float t[3] = {0, 0, 0};
        fn(t, false);                       // warm-up (also the checked run)
        CK(hipDeviceSynchronize());
        const bool ok = !check || verify(c, B, name);
        const int reps = 5;
        Timer T;
        float whole = 0;
        for (int r = 0; r < reps; r++) { T.start(); fn(t, false); whole += T.stop(); }
        float tk[3] = {0, 0, 0};
        for (int r = 0; r < reps; r++) fn(tk, true);
        whole /= reps;
        printf("%-22s %s  sort %.3f ms  (%.2f TB/s of 2*n*20)   parts %.3f %.3f %.3f\n", name, check ? (ok ? "ok " : "BAD") : "   ",
               whole, 40.0 * c.n / whole / 1e9, tk[0] / reps, tk[1] / reps, tk[2] / reps);
        fflush(stdout);
    };
    run("table 256x16 (h,s,sc)", [&](float *t, bool ti) { sort_table<256, 16>(B, c.n, c.tiles, bits, t, ti); });
    run("table 512x16 (h,s,sc)", [&](float *t, bool ti) { sort_table<512, 16>(B, c.n, c.tiles, bits, t, ti); });
    run("sweep 256x16 (h,ms,sw)", [&](float *t, bool ti) { sort_sweep<256, 16>(B, c.n, c.tiles, bits, t, ti); });
    run("sweep 512x16 (h,ms,sw)", [&](float *t, bool ti) { sort_sweep<512, 16>(B, c.n, c.tiles, bits, t, ti); });
    return 0;
}
