// Workload statistics of one rendered view, computed by the library from the arrays the hot path produced.
// The reference only prints per-stage wall times (train.py:29-60, gaussian.py:45-51); every number the
// benchmark reports (N_visible, D, D*, longest tile list, algorithmic bytes per stage: SURVEY section 8(d))
// comes from here.  Off the hot path: one call = four small reduction kernels + one stream synchronisation.
#include "common.h"

namespace {

constexpr int ST_BLOCKS = 1024;
// device counters: 0 visible, 1 visited (D*), 2 longest list, 3 non-empty tiles, 4 pixels that broke
constexpr int ST_WORDS = 8;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void visible_kernel(int N, const int32_t *__restrict__ radii,
                                                      unsigned long long *__restrict__ words) {
    unsigned long long c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += ST_BLOCKS * 256) c += radii[i] > 0 ? 1ull : 0ull;
    c = wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&words[0], c);
}

__global__ __launch_bounds__(256) void lists_kernel(int tiles, const int2 *__restrict__ ranges,
                                                    unsigned long long *__restrict__ words) {
    int longest = 0;
    unsigned long long nonempty = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < tiles; t += ST_BLOCKS * 256) {
        const int2 r = ranges[t];
        longest = max(longest, r.y - r.x);
        nonempty += r.y > r.x ? 1ull : 0ull;
    }
    longest = wave_max_int(longest);
    nonempty = wave_sum_u64(nonempty);
    if ((threadIdx.x & 63) == 0) {
        if (longest) atomicMax(&words[2], (unsigned long long)longest);
        if (nonempty) atomicAdd(&words[3], nonempty);
    }
}

// one 256-thread block per 16x16 pixel block: the deepest last-contributor of the block is the number of list
// entries its wave visited before every pixel had finished (rasterize.c:99-130 counts per pixel)
__global__ __launch_bounds__(256) void visited_kernel(int W, int H, int blocks_y, const int32_t *__restrict__ n_contrib,
                                                      unsigned long long *__restrict__ words) {
    __shared__ int s_max[4];
    __shared__ unsigned long long s_brk[4];
    const int bx = blockIdx.x / blocks_y, by = blockIdx.x % blocks_y;
    const int px = bx * 16 + (threadIdx.x >> 4), py = by * 16 + (threadIdx.x & 15);
    int cnt = 0;
    unsigned long long broke = 0;
    if (px < W && py < H) {
        const int32_t nc = n_contrib[(size_t)px * H + py];
        cnt = nc & 0x7fffffff;
        broke = nc < 0 ? 1ull : 0ull;
    }
    cnt = wave_max_int(cnt);
    broke = wave_sum_u64(broke);
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = cnt; s_brk[threadIdx.x >> 6] = broke; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        const unsigned long long b = s_brk[0] + s_brk[1] + s_brk[2] + s_brk[3];
        if (m) atomicAdd(&words[1], (unsigned long long)m);
        if (b) atomicAdd(&words[4], b);
    }
}

}  // namespace

extern "C" int gs_view_stats(gs_ctx *ctx, int32_t N, const int32_t *radii, int64_t D, const int32_t *ranges,
                             int32_t res_w, int32_t res_h, int32_t tile_size, const int32_t *n_contrib,
                             gs_stats *out_host) {
    GS_REQUIRE(ctx && out_host, "gs_view_stats: ctx / out is NULL");
    GS_REQUIRE(N >= 0 && D >= 0 && res_w > 0 && res_h > 0 && tile_size > 0, "gs_view_stats: invalid sizes");
    GS_REQUIRE(((uintptr_t)ranges & 7) == 0, "gs_view_stats: ranges must be 8-byte aligned");
    GS_HIP(hipSetDevice(ctx->device));
    int rc = gs_buf_reserve(ctx, &ctx->stats_words, ST_WORDS * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long *words = (unsigned long long *)ctx->stats_words.p;
    GS_HIP(hipMemsetAsync(words, 0, ST_WORDS * sizeof(unsigned long long), ctx->stream));
    const int tiles = gs_div_up(res_w, tile_size) * gs_div_up(res_h, tile_size);
    if (radii && N > 0)
        hipLaunchKernelGGL(visible_kernel, dim3(ST_BLOCKS), dim3(256), 0, ctx->stream, N, radii, words);
    if (ranges)
        hipLaunchKernelGGL(lists_kernel, dim3(ST_BLOCKS), dim3(256), 0, ctx->stream, tiles,
                           reinterpret_cast<const int2 *>(ranges), words);
    const int bx = gs_div_up(res_w, 16), by = gs_div_up(res_h, 16);
    if (n_contrib)
        hipLaunchKernelGGL(visited_kernel, dim3(bx * by), dim3(256), 0, ctx->stream, res_w, res_h, by, n_contrib, words);
    GS_HIP(hipGetLastError());
    unsigned long long host[ST_WORDS];
    GS_HIP(hipMemcpyAsync(host, words, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    gs_stats &s = *out_host;
    s.n_gaussians = N;
    s.n_visible = radii ? (int64_t)host[0] : -1;
    s.duplicates = D;
    s.visited = n_contrib ? (int64_t)host[1] : -1;
    s.max_tile_list = ranges ? (int64_t)host[2] : -1;
    s.nonempty_tiles = ranges ? (int64_t)host[3] : -1;
    s.pixels_broke = n_contrib ? (int64_t)host[4] : -1;
    s.pixels = (int64_t)res_w * res_h;
    s.tiles = tiles;
    // algorithmic bytes per launch (restated for the split depth / tile sort)
    const int64_t P = s.pixels, Dv = s.visited < 0 ? 0 : s.visited;
    const int64_t param_bytes = 4 * (3 + 3 + 4 + 48);                 // 232 + the opacity = 236
    s.bytes_preprocess = (int64_t)N * (param_bytes + 4 + 56);
    s.bytes_depth_sort = 2 * (int64_t)N * 20;
    s.bytes_duplicate = (int64_t)N * 28 + D * 8;
    s.bytes_tile_sort = 2 * D * 20;
    s.bytes_ranges = D * 4 + (int64_t)tiles * 8;
    s.bytes_blend_forward = Dv * 40 + P * 20;
    s.bytes_blend_backward = Dv * 40 + P * 20 + Dv * 36;
    s.bytes_param_backward = (int64_t)N * (param_bytes + 4 + 36) + (int64_t)N * (param_bytes + 4);
    return GS_OK;
}
