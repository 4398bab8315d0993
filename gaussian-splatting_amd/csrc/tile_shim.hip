// rasterize_tile: the reference's only native symbol (rasterize.c:27-28), kept as a drop-in for
// c_rasterize's ctypes binding (rasterization.py:112-139).  HOST pointers in and out; the tile is
// blended on the device by the same kernel as gs_blend_forward (pixel offset +0.5,
// rasterize.c:103-104).  Differences from the original: conics are double (defect B7), no
// file-scope globals (rasterize.c:12-23), errors are reported (<0) instead of always returning 0.
// The signature carries no gaussian count, so only the stack_size referenced gaussians are
// gathered on the host and uploaded (re-indexed 0..stack_size-1).
#include "common.h"
#include <mutex>
#include <vector>

namespace {
gs_ctx *g_shim_ctx = nullptr;
std::mutex g_shim_mu;
}

extern "C" int rasterize_tile(double image_chunk[16][16][3], int last_contributors[16][16],
                              double centers[][2], double colors[][3], double opacities[],
                              double conics[][3], int gaus_stack[], double background_color[3],
                              int stack_size, int training, int tile_x, int tile_y, int result_x,
                              int result_y) {
    (void)training;   // ignored by the original too (rasterize.c:28)
    GS_REQUIRE(image_chunk && last_contributors && background_color, "rasterize_tile: NULL output/background");
    GS_REQUIRE(stack_size >= 0, "rasterize_tile: stack_size %d < 0", stack_size);
    GS_REQUIRE(stack_size == 0 || (centers && colors && opacities && conics && gaus_stack),
               "rasterize_tile: NULL array argument");
    GS_REQUIRE(tile_x >= 0 && tile_y >= 0 && result_x > 0 && result_y > 0 && tile_x * 16 < result_x &&
                   tile_y * 16 < result_y, "rasterize_tile: tile (%d,%d) outside %dx%d", tile_x, tile_y,
               result_x, result_y);
    std::lock_guard<std::mutex> lock(g_shim_mu);
    if (!g_shim_ctx) {
        int dev = 0;
        GS_HIP(hipGetDevice(&dev));
        int rc = gs_ctx_create(dev, nullptr, &g_shim_ctx);
        if (rc) return rc;
    }
    gs_ctx *ctx = g_shim_ctx;
    const int n = stack_size;
    // host staging, fp32, gathered in stack order; the tile is moved to the origin of a 16x16
    // frame so the device buffers stay 16x16 (centres shifted by the tile origin)
    const int cols = ((tile_x + 1) * 16 > result_x) ? result_x - tile_x * 16 : 16;
    const int rows = ((tile_y + 1) * 16 > result_y) ? result_y - tile_y * 16 : 16;
    std::vector<float> h((size_t)n * 9 + 1);
    std::vector<int32_t> hv((size_t)n + 1);
    float *h_c = h.data(), *h_col = h_c + 2 * (size_t)n, *h_o = h_col + 3 * (size_t)n, *h_con = h_o + n;
    for (int i = 0; i < n; i++) {
        const int g = gaus_stack[i];
        GS_REQUIRE(g >= 0, "rasterize_tile: negative gaussian index %d at stack position %d", g, i);
        // shift in double before rounding so no precision is lost on large frames
        h_c[2 * i] = (float)(centers[g][0] - 16.0 * tile_x);
        h_c[2 * i + 1] = (float)(centers[g][1] - 16.0 * tile_y);
        for (int k = 0; k < 3; k++) {
            h_col[3 * i + k] = (float)colors[g][k];
            h_con[3 * i + k] = (float)conics[g][k];
        }
        h_o[i] = (float)opacities[g];
        hv[i] = i;
    }
    const size_t f_bytes = ((size_t)n * 9 + 1) * 4, v_bytes = ((size_t)n + 1) * 4;
    const size_t off_v = (f_bytes + 255) & ~(size_t)255, off_r = (off_v + v_bytes + 255) & ~(size_t)255;
    const size_t off_img = off_r + 256, off_nc = off_img + 16 * 16 * 3 * 4, off_tf = off_nc + 16 * 16 * 4;
    int rc = gs_buf_reserve(ctx, &ctx->tile_scratch, off_tf + 16 * 16 * 4);
    if (rc) return rc;
    char *d = (char *)ctx->tile_scratch.p;
    GS_HIP(hipMemcpyAsync(d, h.data(), f_bytes, hipMemcpyHostToDevice, ctx->stream));
    GS_HIP(hipMemcpyAsync(d + off_v, hv.data(), v_bytes, hipMemcpyHostToDevice, ctx->stream));
    const int32_t range[2] = {0, n};
    GS_HIP(hipMemcpyAsync(d + off_r, range, 8, hipMemcpyHostToDevice, ctx->stream));
    const float bg[3] = {(float)background_color[0], (float)background_color[1], (float)background_color[2]};
    const float *dc = (const float *)d;
    rc = gs_blend_forward(ctx, n, dc, dc + 2 * (size_t)n, dc + 5 * (size_t)n, dc + 6 * (size_t)n,
                          (const int32_t *)(d + off_v), (const int32_t *)(d + off_r), n, bg, cols, rows, 16,
                          0.5f, (float *)(d + off_img), (int32_t *)(d + off_nc), (float *)(d + off_tf));
    if (rc) return rc;
    float img[16 * 16 * 3];
    int32_t nc[16 * 16];
    GS_HIP(hipMemcpyAsync(img, d + off_img, (size_t)cols * rows * 12, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipMemcpyAsync(nc, d + off_nc, (size_t)cols * rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    GS_HIP(hipStreamSynchronize(ctx->stream));
    for (int lx = 0; lx < cols; lx++)
        for (int ly = 0; ly < rows; ly++) {
            const int p = lx * rows + ly;   // device frame is [cols][rows]
            for (int k = 0; k < 3; k++) image_chunk[lx][ly][k] = (double)img[3 * p + k];
            last_contributors[lx][ly] = nc[p] & 0x7fffffff;
        }
    return GS_OK;
}
