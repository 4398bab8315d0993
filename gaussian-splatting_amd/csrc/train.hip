// Training-loop side of the hot path: the loss gradient that feeds K7 and
// the parameter update that consumes K8's output.  Both are HBM-bound element-wise passes.
//
// gs_image_loss   replaces train.py:45-49: the reference computes an L1 "loss" on uint8 copies of
//                 the two images (wrap-around, defect B12) and never forms dL/dimage (its kernel
//                 uses the rendered image itself, defect B4).  Here: loss = mean|image - target|
//                 (L1) or mean (image - target)^2 (L2) over the W*H*3 floats, and
//                 dL_dimage = scale * sign(diff)/n  resp.  scale * 2 diff/n, in one pass.
// gs_adam_step    replaces train.py:51-56: the reference's update is `param -= lr * grad * loss`
//                 with a log() and an operator-precedence slip (B12).  Here: Adam (Kingma & Ba) per
//                 parameter group with the reference's five learning rates (main.py:11-17), or the
//                 reference's intent (mode 1: param -= lr * grad * loss_scale) for comparison.
#include "common.h"

namespace {

constexpr int LOSS_BLOCKS = 1024;

template <int KIND>   // 0 = L1, 1 = L2
__global__ __launch_bounds__(256) void image_loss_kernel(size_t n4, size_t n, const float *__restrict__ image,
                                                         const float *__restrict__ target, float gscale,
                                                         float *__restrict__ dL, double *__restrict__ partial) {
    __shared__ double s_sum[4];
    double acc = 0.0;
    const float4 *im4 = reinterpret_cast<const float4 *>(image), *tg4 = reinterpret_cast<const float4 *>(target);
    float4 *dl4 = reinterpret_cast<float4 *>(dL);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)LOSS_BLOCKS * 256) {
        const float4 a = im4[i], b = tg4[i];
        const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (KIND == 0) {
                acc += fabsf(d[k]);
                g[k] = d[k] > 0.0f ? gscale : (d[k] < 0.0f ? -gscale : 0.0f);
            } else {
                acc += (double)d[k] * d[k];
                g[k] = 2.0f * gscale * d[k];
            }
        }
        dl4[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
    // tail (n not a multiple of 4): block 0 only
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float d = image[i] - target[i];
            if (KIND == 0) {
                acc += fabsf(d);
                dL[i] = d > 0.0f ? gscale : (d < 0.0f ? -gscale : 0.0f);
            } else {
                acc += (double)d * d;
                dL[i] = 2.0f * gscale * d;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// folds the per-block partials: loss_out[0] (+)= sum / n
__global__ __launch_bounds__(LOSS_BLOCKS) void image_loss_total_kernel(const double *__restrict__ partial, double inv_n,
                                                                       int accumulate, float *__restrict__ loss_out) {
    __shared__ double s_sum[LOSS_BLOCKS / 64];
    double v = partial[threadIdx.x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < LOSS_BLOCKS / 64; w++) t += s_sum[w];
        const float l = (float)(t * inv_n);
        loss_out[0] = accumulate ? loss_out[0] + l : l;
    }
}

struct AdamGroups {
    float *param[5];        // means, log_scales, quats, opacity, sh
    long long begin[6];     // element offsets of the groups inside the flat buffers (multiples of 4)
    long long count[5];     // elements in each group (3N, 3N, 4N, N, 48N)
    float lr[5];
};

// one thread per float4 of the flat buffers; groups start at multiples of N, so a float4 may
// straddle two groups: the group is looked up per element
__global__ __launch_bounds__(256) void adam_kernel(long long total, AdamGroups gr, const float *__restrict__ grad,
                                                   float *__restrict__ m, float *__restrict__ v, float beta1,
                                                   float beta2, float eps, float c1, float c2, int mode,
                                                   float sgd_scale) {
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    const bool full = i0 + 3 < total && (((uintptr_t)(grad + i0)) & 15) == 0;
    float g[4], mm[4], vv[4];
    if (full) {
        const float4 g4 = *reinterpret_cast<const float4 *>(grad + i0);
        g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
        if (mode == 0) {
            const float4 m4 = *reinterpret_cast<const float4 *>(m + i0), v4 = *reinterpret_cast<const float4 *>(v + i0);
            mm[0] = m4.x; mm[1] = m4.y; mm[2] = m4.z; mm[3] = m4.w;
            vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
        }
    } else {
        for (int k = 0; k < 4; k++) {
            const bool in = i0 + k < total;
            g[k] = in ? grad[i0 + k] : 0.0f;
            mm[k] = (in && mode == 0) ? m[i0 + k] : 0.0f;
            vv[k] = (in && mode == 0) ? v[i0 + k] : 0.0f;
        }
    }
    int grp = 0;
    while (grp < 4 && i0 >= gr.begin[grp + 1]) grp++;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = i0 + k;
        if (i >= total) break;
        while (grp < 4 && i >= gr.begin[grp + 1]) grp++;
        const long long e = i - gr.begin[grp];
        if (e >= gr.count[grp]) continue;                       // alignment padding between groups
        float *p = gr.param[grp] + e;
        float gk = g[k];
        if (!(gk == gk) || fabsf(gk) > 3.0e38f) gk = 0.0f;      // np.nan_to_num (train.py:52-55)
        if (mode == 0) {
            mm[k] = beta1 * mm[k] + (1.0f - beta1) * gk;
            vv[k] = beta2 * vv[k] + (1.0f - beta2) * gk * gk;
            const float mhat = mm[k] * c1, vhat = vv[k] * c2;   // c1 = 1/(1-beta1^t), c2 = 1/(1-beta2^t)
            *p -= gr.lr[grp] * mhat / (sqrtf(vhat) + eps);
        } else {
            *p -= gr.lr[grp] * gk * sgd_scale;
        }
    }
    if (mode == 0) {
        if (full) {
            *reinterpret_cast<float4 *>(m + i0) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4 *>(v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int k = 0; k < 4 && i0 + k < total; k++) { m[i0 + k] = mm[k]; v[i0 + k] = vv[k]; }
        }
    }
}

}  // namespace

extern "C" int gs_image_loss(gs_ctx *ctx, int64_t n, const float *image, const float *target, int32_t kind,
                             float scale, int32_t accumulate, float *dL_dimage, float *loss_out) {
    GS_REQUIRE(ctx, "gs_image_loss: ctx is NULL");
    GS_REQUIRE(n > 0, "gs_image_loss: n=%lld must be positive", (long long)n);
    GS_REQUIRE(kind == 0 || kind == 1, "gs_image_loss: kind %d (0 = L1, 1 = L2)", kind);
    GS_REQUIRE(image && target && dL_dimage && loss_out, "gs_image_loss: NULL argument");
    GS_REQUIRE((((uintptr_t)image | (uintptr_t)target | (uintptr_t)dL_dimage) & 15) == 0,
               "gs_image_loss: image, target and dL_dimage must be 16-byte aligned");
    GS_HIP(hipSetDevice(ctx->device));
    int rc = gs_buf_reserve(ctx, &ctx->loss_partials, LOSS_BLOCKS * sizeof(double));
    if (rc) return rc;
    double *partial = (double *)ctx->loss_partials.p;
    const float gscale = scale / (float)n;
    {
        gs_prof_scope prof_(ctx, GS_K_LOSS);
        if (kind == 0)
            hipLaunchKernelGGL(image_loss_kernel<0>, dim3(LOSS_BLOCKS), dim3(256), 0, ctx->stream, (size_t)n / 4, (size_t)n,
                               image, target, gscale, dL_dimage, partial);
        else
            hipLaunchKernelGGL(image_loss_kernel<1>, dim3(LOSS_BLOCKS), dim3(256), 0, ctx->stream, (size_t)n / 4, (size_t)n,
                               image, target, gscale, dL_dimage, partial);
        hipLaunchKernelGGL(image_loss_total_kernel, dim3(1), dim3(LOSS_BLOCKS), 0, ctx->stream, (const double *)partial,
                           1.0 / (double)n, accumulate, loss_out);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_adam_step(gs_ctx *ctx, int32_t N, float *means, float *log_scales, float *quats, float *opacity,
                            float *sh, const float *grad_flat, float *m_flat, float *v_flat, const float *lr_host5,
                            float beta1, float beta2, float eps, int32_t step, int32_t mode, float sgd_scale) {
    GS_REQUIRE(ctx, "gs_adam_step: ctx is NULL");
    GS_REQUIRE(N >= 0, "gs_adam_step: N=%d < 0", N);
    GS_REQUIRE(mode == 0 || mode == 1, "gs_adam_step: mode %d (0 = Adam, 1 = lr*grad*scale)", mode);
    GS_REQUIRE(step >= 1, "gs_adam_step: step %d must be >= 1", step);
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && log_scales && quats && opacity && sh && grad_flat && lr_host5, "gs_adam_step: NULL argument");
    GS_REQUIRE(mode == 1 || (m_flat && v_flat), "gs_adam_step: Adam needs the moment buffers");
    GS_REQUIRE(mode == 1 || ((((uintptr_t)m_flat | (uintptr_t)v_flat) & 15) == 0), "gs_adam_step: moments must be 16-byte aligned");
    AdamGroups gr;
    float *ps[5] = {means, log_scales, quats, opacity, sh};
    const long long sizes[5] = {3, 3, 4, 1, 48};
    long long off = 0;
    for (int k = 0; k < 5; k++) {
        gr.param[k] = ps[k];
        gr.begin[k] = off;
        gr.count[k] = sizes[k] * (long long)N;
        gr.lr[k] = lr_host5[k];
        off = (off + sizes[k] * (long long)N + 3) / 4 * 4;     // groups start at multiples of 4 floats
    }
    gr.begin[5] = off;
    const float c1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    const float c2 = (float)(1.0 / (1.0 - pow((double)beta2, (double)step)));
    GS_HIP(hipSetDevice(ctx->device));
    const long long threads = (off + 3) / 4;
    {
        gs_prof_scope prof_(ctx, GS_K_ADAM);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, off, gr, grad_flat,
                           m_flat, v_flat, beta1, beta2, eps, c1, c2, mode, sgd_scale);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}
