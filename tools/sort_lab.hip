// Laboratory for the tile sort (not part of the library): times the radix pass variants of csrc/radix_sort.h on
// synthetic (tile id, gaussian) pairs and checks them against std::stable_sort.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_lab.hip -o tools/sort_lab && tools/sort_lab [D] [tiles]
#include "../gaussian-splatting_amd/csrc/radix_sort.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- old kernels (binning_kernels.h) as the baseline
__global__ __launch_bounds__(RADIX_THREADS) void old_hist(uint32_t n, const uint32_t *keys, int shift, uint32_t mask, uint32_t nblocks, uint32_t *table) {
    radix_hist_body(blockIdx.x, n, keys, shift, mask, nblocks, table);
}
__global__ __launch_bounds__(RADIX_THREADS) void scan_k(uint32_t nblocks, uint32_t *table, uint32_t *totals) {
    radix_scan_body(blockIdx.x, nblocks, table, totals);
}
template <int NBITS>
__global__ __launch_bounds__(RADIX_THREADS) void old_scatter(uint32_t n, const uint32_t *ki, const int32_t *vi, uint32_t *ko, int32_t *vo, int shift, uint32_t mask, uint32_t nblocks, const uint32_t *table, const uint32_t *totals) {
    radix_scatter_body<NBITS>(blockIdx.x, n, ki, vi, ko, vo, shift, mask, nblocks, table, totals);
}
// ---- new kernels
template <int THREADS, int ITEMS, bool PAIRS>
__global__ __launch_bounds__(THREADS) void new_hist(uint32_t n, const uint32_t *keys, const uint2 *pairs, int shift, uint32_t mask, uint32_t nblocks, uint32_t *table) {
    radix2_hist_body<THREADS, ITEMS, PAIRS>(blockIdx.x, n, keys, pairs, shift, mask, nblocks, table);
}
template <int NBITS, int THREADS, int ITEMS, int MODE>
__global__ __launch_bounds__(THREADS) void new_scatter(uint32_t n, RadixIO io, int shift, uint32_t mask, uint32_t nblocks, const uint32_t *table, const uint32_t *totals) {
    radix2_scatter_body<NBITS, THREADS, ITEMS, MODE>(blockIdx.x, n, io, shift, mask, nblocks, table, totals);
}

struct Bufs {
    uint32_t *k[2]; int32_t *v[2]; uint2 *p[2]; uint32_t *table; int32_t *out_vals; uint32_t *out_keys;
};

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    void start() { CK(hipEventRecord(a, 0)); }
    float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

static int bits_of(uint32_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

// one full sort (all passes); returns per-kernel-kind ms in t[3] (hist, scan, scatter)
template <int THREADS, int ITEMS, bool PAIRS>
void sort_new(Bufs &B, uint32_t n, int bits, float t[3], bool timeit) {
    constexpr int TILE = THREADS * ITEMS;
    const uint32_t nblocks = (n + TILE - 1) / TILE;
    uint32_t *totals = B.table, *table = B.table + RADIX_BINS;
    const int passes = (bits + 7) / 8, per = (bits + passes - 1) / passes;
    Timer T;
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        const int shift = p * per, nb = std::min(per, bits - shift);
        const uint32_t mask = (1u << nb) - 1u;
        const bool last = p == passes - 1;
        RadixIO io{};
        io.keys_in = B.k[cur]; io.vals_in = B.v[cur]; io.pairs_in = B.p[cur];
        io.keys_out = last ? B.out_keys : B.k[cur ^ 1]; io.vals_out = last ? B.out_vals : B.v[cur ^ 1]; io.pairs_out = B.p[cur ^ 1];
        if (timeit) T.start();
        hipLaunchKernelGGL((new_hist<THREADS, ITEMS, PAIRS>), dim3(nblocks), dim3(THREADS), 0, 0, n, io.keys_in, io.pairs_in, shift, mask, nblocks, table);
        if (timeit) t[0] += T.stop();
        if (timeit) T.start();
        hipLaunchKernelGGL(scan_k, dim3(RADIX_BINS), dim3(RADIX_THREADS), 0, 0, nblocks, table, totals);
        if (timeit) t[1] += T.stop();
        if (timeit) T.start();
#define LAUNCH(NB, MODE) hipLaunchKernelGGL((new_scatter<NB, THREADS, ITEMS, MODE>), dim3(nblocks), dim3(THREADS), 0, 0, n, io, shift, mask, nblocks, (const uint32_t *)table, (const uint32_t *)totals)
#define BY_MODE(NB) do { if (!PAIRS) LAUNCH(NB, RADIX_SOA); else if (last) LAUNCH(NB, RADIX_PAIRS_TO_SOA); else LAUNCH(NB, RADIX_PAIRS); } while (0)
        switch (nb) { case 5: BY_MODE(5); break; case 6: BY_MODE(6); break; case 7: BY_MODE(7); break; case 8: BY_MODE(8); break; default: printf("nbits %d not instantiated\n", nb); exit(1); }
        if (timeit) t[2] += T.stop();
        cur ^= 1;
    }
    CK(hipGetLastError());
}

void sort_old(Bufs &B, uint32_t n, int bits, float t[3], bool timeit) {
    const uint32_t nblocks = (n + RADIX_TILE - 1) / RADIX_TILE;
    uint32_t *totals = B.table, *table = B.table + RADIX_BINS;
    const int passes = (bits + 7) / 8, per = (bits + passes - 1) / passes;
    Timer T;
    int cur = 0;
    for (int p = 0; p < passes; p++) {
        const int shift = p * per, nb = std::min(per, bits - shift);
        const uint32_t mask = (1u << nb) - 1u;
        const bool last = p == passes - 1;
        uint32_t *ko = last ? B.out_keys : B.k[cur ^ 1];
        int32_t *vo = last ? B.out_vals : B.v[cur ^ 1];
        if (timeit) T.start();
        hipLaunchKernelGGL(old_hist, dim3(nblocks), dim3(RADIX_THREADS), 0, 0, n, B.k[cur], shift, mask, nblocks, table);
        if (timeit) t[0] += T.stop();
        if (timeit) T.start();
        hipLaunchKernelGGL(scan_k, dim3(RADIX_BINS), dim3(RADIX_THREADS), 0, 0, nblocks, table, totals);
        if (timeit) t[1] += T.stop();
        if (timeit) T.start();
#define OLD(NB) hipLaunchKernelGGL(old_scatter<NB>, dim3(nblocks), dim3(RADIX_THREADS), 0, 0, n, (const uint32_t *)B.k[cur], (const int32_t *)B.v[cur], ko, vo, shift, mask, nblocks, (const uint32_t *)table, (const uint32_t *)totals)
        switch (nb) { case 5: OLD(5); break; case 6: OLD(6); break; case 7: OLD(7); break; case 8: OLD(8); break; default: exit(1); }
        if (timeit) t[2] += T.stop();
        cur ^= 1;
    }
    CK(hipGetLastError());
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 122469486u;
    const uint32_t tiles = argc > 2 ? (uint32_t)atoi(argv[2]) : 32400u;
    const int bits = bits_of(tiles - 1);
    printf("n=%u tiles=%u bits=%d  algorithmic bytes 2*n*20 = %.1f MB\n", n, tiles, bits, 40.0 * n / 1e6);
    std::vector<uint32_t> hk(n);
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (uint32_t)(s % tiles); }
    std::vector<uint2> hp(n);
    for (uint32_t i = 0; i < n; i++) hp[i] = make_uint2(hk[i], i);
    Bufs B{};
    const size_t n4 = ((size_t)n * 4 + 255) & ~(size_t)255;
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&B.k[i], n4)); CK(hipMalloc(&B.v[i], n4)); CK(hipMalloc(&B.p[i], 2 * n4)); }
    CK(hipMalloc(&B.out_vals, n4)); CK(hipMalloc(&B.out_keys, n4));
    CK(hipMalloc(&B.table, ((size_t)RADIX_BINS * (n / 4096 + 2) + RADIX_BINS) * 4));
    std::vector<int32_t> iota(n);
    std::iota(iota.begin(), iota.end(), 0);
    std::vector<int32_t> want;
    const bool check = n <= 8000000u;
    if (check) {
        want = iota;
        std::stable_sort(want.begin(), want.end(), [&](int32_t a, int32_t b) { return hk[a] < hk[b]; });
    }
    auto reset = [&]() {
        CK(hipMemcpy(B.k[0], hk.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.v[0], iota.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.p[0], hp.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    };
    auto run = [&](const char *name, auto fn) {
        reset();
        float t[3] = {0, 0, 0};
        fn(t, false);                       // warm-up (also the checked run)
        CK(hipDeviceSynchronize());
        bool ok = true;
        if (check) {
            std::vector<int32_t> got(n);
            std::vector<uint32_t> gk(n);
            CK(hipMemcpy(got.data(), B.out_vals, (size_t)n * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gk.data(), B.out_keys, (size_t)n * 4, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; i++) if (got[i] != want[i] || gk[i] != hk[want[i]]) { ok = false; printf("  MISMATCH at %u: got %d want %d\n", i, got[i], want[i]); break; }
        }
        const int reps = 5;
        Timer T;
        float whole = 0;
        for (int r = 0; r < reps; r++) { T.start(); fn(t, false); whole += T.stop(); }
        float tk[3] = {0, 0, 0};
        for (int r = 0; r < reps; r++) fn(tk, true);
        whole /= reps;
        printf("%-28s %s  sort %.3f ms  (%.2f TB/s of 2*n*20)   hist %.3f scan %.3f scatter %.3f\n", name, check ? (ok ? "ok " : "BAD") : "   ",
               whole, 40.0 * n / whole / 1e9, tk[0] / reps, tk[1] / reps, tk[2] / reps);
        fflush(stdout);
    };
    run("old 256x16 soa", [&](float *t, bool ti) { sort_old(B, n, bits, t, ti); });
    run("new 256x16 soa", [&](float *t, bool ti) { sort_new<256, 16, false>(B, n, bits, t, ti); });
    run("new 512x16 soa", [&](float *t, bool ti) { sort_new<512, 16, false>(B, n, bits, t, ti); });
    run("new 1024x16 soa", [&](float *t, bool ti) { sort_new<1024, 16, false>(B, n, bits, t, ti); });
    run("new 256x16 pairs", [&](float *t, bool ti) { sort_new<256, 16, true>(B, n, bits, t, ti); });
    run("new 512x16 pairs", [&](float *t, bool ti) { sort_new<512, 16, true>(B, n, bits, t, ti); });
    run("new 1024x16 pairs", [&](float *t, bool ti) { sort_new<1024, 16, true>(B, n, bits, t, ti); });
    run("new 1024x8 pairs", [&](float *t, bool ti) { sort_new<1024, 8, true>(B, n, bits, t, ti); });
    run("new 512x8 pairs", [&](float *t, bool ti) { sort_new<512, 8, true>(B, n, bits, t, ti); });
    return 0;
}
