// K1 preprocess: replaces forward_pass (forward_pass.py:25-136) + py_compute_color (:168-215).
//
// One thread per gaussian.  HBM-bound (236 B in, 56 B out per gaussian, +120 B of training
// extras when requested), so the whole chain is evaluated in fp64 at no cost: the discontinuous
// outputs -- depth key int(z*1000) (:70-71), radius ceil(3*sqrt(lambda)) (:106-112) and the tile
// rect int(clamp(.)/tile) (:115-120) -- then equal the reference's fp64 values.  This file is
// compiled with -ffp-contract=off so no FMA contraction changes those roundings.
#include "common.h"
#include "precam.h"
#include <math.h>

namespace {

// int(min(size, max(0, v)) / tile) with Python's min/max semantics (forward_pass.py:116-119)
__device__ inline int clamp_to_tile(double v, double size, double tile) {
    double m = (v > 0.0) ? v : 0.0;
    m = (m < size) ? m : size;
    return (int)(m / tile);
}

__global__ __launch_bounds__(256) void preprocess_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ log_scales,
    const float *__restrict__ quats, const float *__restrict__ sh, PreCam cam,
    float *__restrict__ centers, int32_t *__restrict__ depths, float *__restrict__ colors,
    float *__restrict__ conics, uint8_t *__restrict__ clamped, int32_t *__restrict__ tiles_touched,
    int32_t *__restrict__ radii, float *__restrict__ Ws, float *__restrict__ Ms,
    float *__restrict__ Ts, float *__restrict__ cov2d, float *__restrict__ cov3d,
    float *__restrict__ dirs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;

    // outputs of a culled gaussian are all zero (the reference pre-zeroes, forward_pass.py:29-41)
    float o_cx = 0.f, o_cy = 0.f, o_col[3] = {0.f, 0.f, 0.f}, o_con[3] = {0.f, 0.f, 0.f};
    uint8_t o_cl[3] = {0, 0, 0};
    int o_depth = 0, o_rad = 0, o_rect[4] = {0, 0, 0, 0};
    double eW[6] = {0, 0, 0, 0, 0, 0}, eM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, eT[3] = {0, 0, 0},
           eC2[3] = {0, 0, 0}, eC3[6] = {0, 0, 0, 0, 0, 0}, eD[3] = {0, 0, 0};

    const double x = means[3 * i], y = means[3 * i + 1], z = means[3 * i + 2];

    do {
        // :59-61  clip-space centre from WORLD coordinates (quirk Q1)
        const double mod_w = 1.0 / (z + .00000001);
        const double ndc_x = (cam.p00 * x) * mod_w;
        const double ndc_y = (cam.p11 * y) * mod_w;

        // :64-67  view-space position, near cull at z <= 2 (quirk Q6)
        const double tx = ((cam.Rt[0] * x + cam.Rt[1] * y) + cam.Rt[2] * z) + cam.t[0];
        const double ty = ((cam.Rt[3] * x + cam.Rt[4] * y) + cam.Rt[5] * z) + cam.t[1];
        const double tz = ((cam.Rt[6] * x + cam.Rt[7] * y) + cam.Rt[8] * z) + cam.t[2];
        if (tz <= 2.0 || isnan(tz)) break;

        // :70-71  depth key = int(z*1000) wrapped to int32 (quirk Q7)
        {
            const double d = trunc(tz * 1000.0);
            if (d >= -2147483648.0 && d < 2147483648.0) {
                o_depth = (int)d;
            } else if (isfinite(d)) {
                const double w = fmod(d + 2147483648.0, 4294967296.0);
                o_depth = (int)((w < 0 ? w + 4294967296.0 : w) - 2147483648.0);
            }
        }

        // :74  rotation from the normalised quaternion (w,x,y,z); as_rotation_matrix divides by
        // the squared norm once more
        const double q0 = quats[4 * i], q1 = quats[4 * i + 1], q2 = quats[4 * i + 2],
                     q3 = quats[4 * i + 3];
        const double qn = sqrt(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
        const double qw = q0 / qn, qx = q1 / qn, qy = q2 / qn, qz = q3 / qn;
        const double s = 2.0 / (((qw * qw + qx * qx) + qy * qy) + qz * qz);
        const double R[9] = {1 - s * (qy * qy + qz * qz), s * (qx * qy - qz * qw), s * (qx * qz + qy * qw),
                             s * (qx * qy + qz * qw), 1 - s * (qx * qx + qz * qz), s * (qy * qz - qx * qw),
                             s * (qx * qz - qy * qw), s * (qy * qz + qx * qw), 1 - s * (qx * qx + qy * qy)};

        // :75  M = R @ ([[s0,1,1],[1,s1,1],[1,1,s2]] * scale_mod) (quirk Q4)
        const double sm = cam.scale_mod;
        const double S[9] = {exp((double)log_scales[3 * i]) * sm, sm, sm,
                             sm, exp((double)log_scales[3 * i + 1]) * sm, sm,
                             sm, sm, exp((double)log_scales[3 * i + 2]) * sm};
        double M[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                M[3 * r + c] = (R[3 * r] * S[c] + R[3 * r + 1] * S[3 + c]) + R[3 * r + 2] * S[6 + c];
        // :76  Sigma = M M^T
        double Sg[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                Sg[3 * r + c] = (M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1]) + M[3 * r + 2] * M[3 * c + 2];

        // :78-82  W = J^T @ camera_r^T; only rows 0,1 reach cov_2d[0:2,0:2] (quirk Q5)
        const double j00 = cam.fx / tz, j11 = cam.fy / tz;
        const double W0[3] = {j00 * cam.Rt[0], j00 * cam.Rt[1], j00 * cam.Rt[2]};
        const double W1[3] = {j11 * cam.Rt[3], j11 * cam.Rt[4], j11 * cam.Rt[5]};
        double u0[3], u1[3];   // Sigma @ W0, Sigma @ W1
#pragma unroll
        for (int r = 0; r < 3; r++) {
            u0[r] = (Sg[3 * r] * W0[0] + Sg[3 * r + 1] * W0[1]) + Sg[3 * r + 2] * W0[2];
            u1[r] = (Sg[3 * r] * W1[0] + Sg[3 * r + 1] * W1[1]) + Sg[3 * r + 2] * W1[2];
        }
        // :85-86  dilation
        const double a = ((W0[0] * u0[0] + W0[1] * u0[1]) + W0[2] * u0[2]) + .3;
        const double b = (W0[0] * u1[0] + W0[1] * u1[1]) + W0[2] * u1[2];
        const double c = ((W1[0] * u1[0] + W1[1] * u1[1]) + W1[2] * u1[2]) + .3;

        // :89-94  training extras
        for (int k = 0; k < 3; k++) { eW[k] = W0[k]; eW[3 + k] = W1[k]; }
        for (int k = 0; k < 9; k++) eM[k] = M[k];
        eT[0] = tx; eT[1] = ty; eT[2] = tz;
        eC2[0] = a; eC2[1] = b; eC2[2] = c;
        eC3[0] = Sg[0]; eC3[1] = Sg[1]; eC3[2] = Sg[2]; eC3[3] = Sg[4]; eC3[4] = Sg[5]; eC3[5] = Sg[8];

        // :97-103  conic
        const double det = a * c - b * b;
        if (det == 0.0) break;
        const double det_inv = 1.0 / det;
        o_con[0] = (float)(c * det_inv);
        o_con[1] = (float)(-b * det_inv);
        o_con[2] = (float)(a * det_inv);

        // :106-112  radius (quirk Q8)
        const double mid = .5 * (a + c);
        const double disc = sqrt(fmax(.1, mid * mid - det));
        const double l1 = mid + disc, l2 = mid - disc;
        const double val = sqrt(fmax(fmax(l1, l2), 0.0));
        if (isnan(val) || isnan(l1)) break;
        const double radius = ceil(3.0 * val);

        // :114-120  pixel centre (functions.py:38-39) and tile rect (quirk Q9); written before
        // the off-screen cull (quirk Q10)
        const double cx = ((ndc_x + 1.0) * cam.W - 1.0) * .5;
        const double cy = ((ndc_y + 1.0) * cam.H - 1.0) * .5;
        o_rect[0] = clamp_to_tile(cx - radius, cam.W, cam.tile);
        o_rect[1] = clamp_to_tile(cy - radius, cam.H, cam.tile);
        o_rect[2] = clamp_to_tile(cx + radius + cam.tile - 1, cam.W, cam.tile);
        o_rect[3] = clamp_to_tile(cy + radius + cam.tile - 1, cam.H, cam.tile);

        // :122-123  off-screen cull
        if (cx + radius < 0 || cx - radius > cam.W || cy + radius < 0 || cy - radius > cam.H) break;

        o_rad = (int)radius;
        o_cx = (float)cx;
        o_cy = (float)cy;

        // :127 -> py_compute_color (:168-215)
        double dx = x - cam.cam_pos[0], dy = y - cam.cam_pos[1], dz = z - cam.cam_pos[2];
        const double dn = sqrt((dx * dx + dy * dy) + dz * dz);
        dx /= dn; dy /= dn; dz /= dn;
        eD[0] = dx; eD[1] = dy; eD[2] = dz;
        double basis[16];
        basis[0] = SH_C0;
        if (cam.degree > 0) {
            basis[1] = -dy * SH_C1; basis[2] = dz * SH_C1; basis[3] = -dx * SH_C1;
            if (cam.degree > 1) {
                const double xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                basis[4] = SH_C2_0 * xy; basis[5] = SH_C2_1 * yz;
                basis[6] = SH_C2_2 * (2.0 * zz - xx - yy);
                basis[7] = SH_C2_3 * xz; basis[8] = SH_C2_4 * (xx - yy);
                if (cam.degree > 2) {
                    basis[9] = SH_C3_0 * dy * (3.0 * xx - yy);
                    basis[10] = SH_C3_1 * xy * dz;
                    basis[11] = SH_C3_2 * dy * (4.0 * zz - xx - yy);
                    basis[12] = SH_C3_3 * dz * (2.0 * zz - 3.0 * xx - 3.0 * yy);
                    basis[13] = SH_C3_4 * dx * (4.0 * zz - xx - yy);
                    basis[14] = SH_C3_5 * dz * (xx - yy);
                    basis[15] = SH_C3_6 * dx * (xx - 3.0 * yy);
                }
            }
        }
        const int nb = (cam.degree + 1) * (cam.degree + 1);
        const float4 *sh4 = reinterpret_cast<const float4 *>(sh + (size_t)i * 48);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float coef[16];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float4 f = sh4[ch * 4 + v];
                coef[4 * v] = f.x; coef[4 * v + 1] = f.y; coef[4 * v + 2] = f.z; coef[4 * v + 3] = f.w;
            }
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) r += basis[k] * (double)coef[k];
            r += .5;
            if (r < 0) { r = 0; o_cl[ch] = 1; }
            o_col[ch] = (float)r;
        }
    } while (0);

    centers[2 * i] = o_cx; centers[2 * i + 1] = o_cy;
    depths[i] = o_depth;
    radii[i] = o_rad;
    reinterpret_cast<int4 *>(tiles_touched)[i] = make_int4(o_rect[0], o_rect[1], o_rect[2], o_rect[3]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        colors[3 * i + k] = o_col[k];
        conics[3 * i + k] = o_con[k];
        clamped[3 * i + k] = o_cl[k];
    }
    if (Ws) for (int k = 0; k < 6; k++) Ws[6 * i + k] = (float)eW[k];
    if (Ms) for (int k = 0; k < 9; k++) Ms[9 * i + k] = (float)eM[k];
    if (Ts) for (int k = 0; k < 3; k++) Ts[3 * i + k] = (float)eT[k];
    if (cov2d) for (int k = 0; k < 3; k++) cov2d[3 * i + k] = (float)eC2[k];
    if (cov3d) for (int k = 0; k < 6; k++) cov3d[6 * i + k] = (float)eC3[k];
    if (dirs) for (int k = 0; k < 3; k++) dirs[3 * i + k] = (float)eD[k];
}

}  // namespace

// host-side derivation of the per-frame constants with the reference's own formulas
int gs_make_precam(const gs_camera *cam, PreCam *out) {
    PreCam &pc = *out;
    const double fov_x = 2.0 * atan(cam->cam_width / (2.0 * cam->fx));    // functions.py:36-37
    const double fov_y = 2.0 * atan(cam->cam_height / (2.0 * cam->fx));   // fx for both (quirk Q2)
    const double z_near = .01;
    const double top = tan(fov_y / 2) * z_near, right = tan(fov_x / 2) * z_near;
    pc.p00 = z_near / right;
    pc.p11 = z_near / top;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) pc.Rt[3 * r + c] = cam->R[3 * c + r];
    for (int k = 0; k < 3; k++) pc.t[k] = cam->t[k];
    for (int r = 0; r < 3; r++)
        pc.cam_pos[r] = (pc.Rt[3 * r] * cam->t[0] + pc.Rt[3 * r + 1] * cam->t[1]) + pc.Rt[3 * r + 2] * cam->t[2];
    pc.fx = cam->fx;
    pc.fy = cam->fy;
    pc.scale_mod = (1.0 / cam->res_h) * 3;
    pc.W = cam->res_w; pc.H = cam->res_h; pc.tile = cam->tile_size; pc.degree = cam->sh_degree;
    return GS_OK;
}

extern "C" int gs_preprocess(gs_ctx *ctx, int32_t N, const float *means, const float *log_scales,
                             const float *quats, const float *sh, const gs_camera *cam,
                             float *centers, int32_t *depths, float *colors, float *conics,
                             uint8_t *clamped, int32_t *tiles_touched, int32_t *radii,
                             float *Ws, float *Ms, float *Ts, float *cov2d, float *cov3d,
                             float *dirs) {
    GS_REQUIRE(ctx && cam, "gs_preprocess: ctx/cam is NULL");
    GS_REQUIRE(N >= 0, "gs_preprocess: N=%d < 0", N);
    GS_REQUIRE(cam->sh_degree >= 0 && cam->sh_degree <= 3, "gs_preprocess: sh_degree %d not in 0..3",
               cam->sh_degree);
    GS_REQUIRE(cam->res_w > 0 && cam->res_h > 0 && cam->tile_size > 0 && cam->cam_width > 0 &&
                   cam->cam_height > 0 && cam->fx > 0 && cam->fy > 0,
               "gs_preprocess: invalid camera/frame description");
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && log_scales && quats && sh && centers && depths && colors && conics &&
                   clamped && tiles_touched && radii,
               "gs_preprocess: NULL array argument");
    GS_REQUIRE(((uintptr_t)sh & 15) == 0 && ((uintptr_t)tiles_touched & 15) == 0,
               "gs_preprocess: sh and tiles_touched must be 16-byte aligned");
    PreCam pc;
    gs_make_precam(cam, &pc);
    GS_HIP(hipSetDevice(ctx->device));
    { gs_prof_scope prof_(ctx, GS_K_PREPROCESS); hipLaunchKernelGGL(preprocess_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                       means, log_scales, quats, sh, pc, centers, depths, colors, conics, clamped,
                       tiles_touched, radii, Ws, Ms, Ts, cov2d, cov3d, dirs); }
    GS_HIP(hipGetLastError());
    return GS_OK;
}
