// K8 per-gaussian parameter backward: replaces backward_pass (backward_pass.py:19-68),
// cov_2d_backpass + kernel cov_2d_grads (backward_pass.py:99-133, setup_gpu.py:168-253),
// sh_backpass + kernel sh_grads/dnormvdv (backward_pass.py:70-97, setup_gpu.py:256-303) and the
// host d_scales/d_rots loop (backward_pass.py:40-66).
//
// The reference's versions are undefined at HEAD (defects B3, B9, B10, B13), so this kernel
// implements the evident intent: the exact derivative of the LIVE forward path
// (forward_pass.py:56-127 with quirks Q1-Q5) w.r.t. the stored parameters, pinned by fp64 autograd
// of the oracle restatement.  One thread per gaussian; the forward intermediates (R, M, Sigma, W
// rows, cov2d) are recomputed in fp64 from the parameters rather than re-read (132 B of extras in
// the reference).
//
// Multi-view batches (the reference trains one view per iteration, train.py:18-43): one launch
// takes up to 8 views.  R, M and Sigma do not depend on the view, dL/dSigma is linear, so the
// thread reads its 236 B of parameters once, loops over the views accumulating dL/dmean, dL/dSigma
// and dL/dSH in registers, runs the Sigma -> (scale, quaternion) chain once and writes its 236 B
// of gradients once: 236 + 8*(64+7) + 236 B per gaussian instead of 8 * (236 + 36 + 2*236) with
// one accumulate-launch per view.
#include "common.h"
#include "precam.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int MAX_BATCH_VIEWS = 8;

struct ViewIn {
    PreCam cam;
    const int32_t *radii;
    const uint8_t *clamped;
    const float *rows;        // ROWS: [N][16] blend-backward rows (d centre 2, d conic 3, d opacity, d colour 3)
    const float *d_centers;   // !ROWS: the reference-shaped separate arrays
    const float *d_conics;
    const float *d_colors;
};

struct BatchArgs {
    ViewIn v[MAX_BATCH_VIEWS];
    int n_views;
};

#ifndef GS_K8_WAVES
#define GS_K8_WAVES 2
#endif
// FAST (chosen per launch on the host): every view has SH degree 3 and the strict clip-space centre, so the degree
// tests of the 48-term sums fold away (a select per term otherwise); same formulas, results equal to fp32 rounding
// (FMA contraction is on in this file and differs between the two forms).
template <bool ROWS, bool FAST>
__global__ __launch_bounds__(256, GS_K8_WAVES) void param_bwd_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ log_scales,
    const float *__restrict__ quats, const float *__restrict__ sh, const float *__restrict__ opacity,
    int opacity_is_logit, const BatchArgs args, int accumulate, float *__restrict__ dL_dmeans,
    float *__restrict__ dL_dlog_scales, float *__restrict__ dL_dquats, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dsh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;

    double g_mean[3] = {0, 0, 0}, g_ls[3] = {0, 0, 0}, g_q[4] = {0, 0, 0, 0};
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // sum over views of G + G^T, G = dL/dSigma
    float g_op = 0.0f;
    float acc_sh[48];
#pragma unroll
    for (int k = 0; k < 48; k++) acc_sh[k] = 0.0f;
    float4 *out_sh = reinterpret_cast<float4 *>(dL_dsh + (size_t)i * 48);
    bool any = false;

    const double x = means[3 * i], y = means[3 * i + 1], z = means[3 * i + 2];
    const double qw = quats[4 * i], qx = quats[4 * i + 1], qy = quats[4 * i + 2], qz = quats[4 * i + 3];
    const double n2 = qw * qw + qx * qx + qy * qy + qz * qz;
    const double s = 2.0 / n2;
    const double Q[9] = {-(qy * qy + qz * qz), qx * qy - qz * qw, qx * qz + qy * qw,
                         qx * qy + qz * qw, -(qx * qx + qz * qz), qy * qz - qx * qw,
                         qx * qz - qy * qw, qy * qz + qx * qw, -(qx * qx + qy * qy)};
    double R[9];
    for (int k = 0; k < 9; k++) R[k] = s * Q[k];
    R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
    const double sc[3] = {exp((double)log_scales[3 * i]), exp((double)log_scales[3 * i + 1]),
                          exp((double)log_scales[3 * i + 2])};

    // The thread's 48 SH coefficients are needed once per view: a thread-private copy in LDS (no
    // barrier: nobody else reads it) instead of 12 global loads per view that miss L1 at this
    // occupancy.  [vv][thread] float4: conflict-free for both the write and the reads.
    // (pitch 261 float4: the transposed reads of the gradient write-out below hit distinct banks)
    constexpr int SH_PITCH = 261;
    __shared__ float4 s_sh4[12][SH_PITCH];
    {
        const float4 *sh4 = reinterpret_cast<const float4 *>(sh + (size_t)i * 48);
#pragma unroll
        for (int vv = 0; vv < 12; vv++) s_sh4[vv][threadIdx.x] = sh4[vv];
    }

    // S', M = R S' and Sigma = M M^T do not depend on the view (scale_mod is a frame constant,
    // checked on the host): once per gaussian (forward_pass.py:74-76)
    const double sm = args.v[0].cam.scale_mod;
    const double so = args.v[0].cam.diag_scale ? 0.0 : sm;     // GS_CAM_DIAG_SCALE (the same for every view: host check)
    const double S[9] = {sc[0] * sm, so, so, so, sc[1] * sm, so, so, so, sc[2] * sm};
    double M[9], Sg[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            M[3 * r + c] = R[3 * r] * S[c] + R[3 * r + 1] * S[3 + c] + R[3 * r + 2] * S[6 + c];
    for (int r = 0; r < 3; r++)
        for (int c = r; c < 3; c++)
            Sg[3 * r + c] = Sg[3 * c + r] = M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1] + M[3 * r + 2] * M[3 * c + 2];

    // per-view inputs, fetched one view ahead of their use (two HBM round trips per view otherwise).  (Staging a wave's
    // 64 rows per view through LDS with three whole-row LDS-DMA loads instead of these register loads of quarter
    // lines was measured: 0.266 against 0.230 ms per 8 views -- the wait for the DMA sits in front of every view.)
    struct Inputs {
        int radius;
        float c[2], con[3], col[3], d_op;
        bool clamp[3];
    };
    auto fetch = [&](int v) {
        Inputs in;
        const ViewIn &V = args.v[v];
        in.radius = V.radii[i];
        in.d_op = 0.0f;
        if (ROWS) {
            const float4 *r = reinterpret_cast<const float4 *>(V.rows + (size_t)i * 16);
            const float4 r0 = r[0], r1 = r[1];
            in.c[0] = r0.x; in.c[1] = r0.y;
            in.con[0] = r0.z; in.con[1] = r0.w; in.con[2] = r1.x;
            in.d_op = r1.y;
            in.col[0] = r1.z; in.col[1] = r1.w; in.col[2] = V.rows[(size_t)i * 16 + 8];
        } else {
            in.c[0] = V.d_centers[2 * i]; in.c[1] = V.d_centers[2 * i + 1];
            for (int k = 0; k < 3; k++) { in.con[k] = V.d_conics[3 * i + k]; in.col[k] = V.d_colors[3 * i + k]; }
        }
        for (int k = 0; k < 3; k++) in.clamp[k] = V.clamped[3 * i + k] != 0;
        return in;
    };
    Inputs nxt = {};
    if (args.n_views > 0) nxt = fetch(0);      // an empty batch only writes zeros (a rank without views)

#pragma unroll 1
    for (int v = 0; v < args.n_views; v++) {
        const PreCam &cam = args.v[v].cam;
#ifdef GS_K8_NOLOAD
        Inputs in = nxt;                       // experiment: view 0's inputs for every view (no per-view loads)
        in.c[0] += (float)v;
#else
        const Inputs in = nxt;
#endif
        const int degree = FAST ? 3 : cam.degree;
#ifndef GS_K8_NOLOAD
        if (v + 1 < args.n_views) nxt = fetch(v + 1);
#endif
        if (in.radius <= 0) continue;
        any = true;
        g_op += in.d_op;
        const float *in_c = in.c, *in_con = in.con, *in_col = in.col;

        // ---- recompute the W rows and cov2d (forward_pass.py:78-86)
        const double tz = cam.Rt[6] * x + cam.Rt[7] * y + cam.Rt[8] * z + cam.t[2];

        // ---- 2D centre <- world mean (quirk Q1: proj_mat on world coordinates)
        if (FAST || !cam.apply_view) {
            const double dcx = in_c[0], dcy = in_c[1];
            const double mw = 1.0 / (z + .00000001);
            g_mean[0] += dcx * (.5 * cam.W * cam.p00 * mw);
            g_mean[1] += dcy * (.5 * cam.H * cam.p11 * mw);
            g_mean[2] += -dcx * (.5 * cam.W * cam.p00 * x * mw * mw) - dcy * (.5 * cam.H * cam.p11 * y * mw * mw);
        } else {    // GS_CAM_APPLY_VIEW: the centre is projected from t = R^T p + t_cam
            const double tx = cam.Rt[0] * x + cam.Rt[1] * y + cam.Rt[2] * z + cam.t[0];
            const double ty = cam.Rt[3] * x + cam.Rt[4] * y + cam.Rt[5] * z + cam.t[1];
            const double mw = 1.0 / (tz + .00000001);
            const double d_tx = in_c[0] * (.5 * cam.W * cam.p00 * mw), d_ty = in_c[1] * (.5 * cam.H * cam.p11 * mw);
            const double d_tzc = -(d_tx * tx + d_ty * ty) * mw;
            for (int k = 0; k < 3; k++) g_mean[k] += d_tx * cam.Rt[k] + d_ty * cam.Rt[3 + k] + d_tzc * cam.Rt[6 + k];
        }

        const double tz_inv = 1.0 / tz;
        const double j00 = cam.fx * tz_inv, j11 = cam.fy * tz_inv;
        const double W0[3] = {j00 * cam.Rt[0], j00 * cam.Rt[1], j00 * cam.Rt[2]};
        const double W1[3] = {j11 * cam.Rt[3], j11 * cam.Rt[4], j11 * cam.Rt[5]};
        double u0[3], u1[3];
        for (int r = 0; r < 3; r++) {
            u0[r] = Sg[3 * r] * W0[0] + Sg[3 * r + 1] * W0[1] + Sg[3 * r + 2] * W0[2];
            u1[r] = Sg[3 * r] * W1[0] + Sg[3 * r + 1] * W1[1] + Sg[3 * r + 2] * W1[2];
        }
        const double a0 = W0[0] * u0[0] + W0[1] * u0[1] + W0[2] * u0[2];
        const double b = W0[0] * u1[0] + W0[1] * u1[1] + W0[2] * u1[2];
        const double c0 = W1[0] * u1[0] + W1[1] * u1[1] + W1[2] * u1[2];
        const double a = a0 + .3, c = c0 + .3;
        const double det = a * c - b * b;

        // ---- conic (A,B,C) = (c, -b, a)/det  ->  cov2d (a,b,c)
        const double gA = in_con[0], gB = in_con[1], gC = in_con[2];
        const double inv2 = 1.0 / (det * det);
        const double d_a = (-c * c * gA + b * c * gB - b * b * gC) * inv2;
        const double d_b = (2.0 * b * c * gA - (det + 2.0 * b * b) * gB + 2.0 * a * b * gC) * inv2;
        const double d_c = (-b * b * gA + a * b * gB - a * a * gC) * inv2;

        // ---- cov2d -> tz (through the 1/tz of both W rows) -> mean
        {
            const double d_tz = -2.0 * (d_a * a0 + d_b * b + d_c * c0) * tz_inv;
            g_mean[0] += d_tz * cam.Rt[6];
            g_mean[1] += d_tz * cam.Rt[7];
            g_mean[2] += d_tz * cam.Rt[8];
        }

        // ---- cov2d -> Sigma: G + G^T, summed over the views (Sigma does not depend on the view)
        // (H is symmetric: six entries, H_rc = 2 d_a W0r W0c + d_b (W0r W1c + W1r W0c) + 2 d_c W1r W1c)
        {
            double aW0[3], bW0[3], bW1[3], cW1[3];
            for (int r = 0; r < 3; r++) {
                aW0[r] = 2.0 * d_a * W0[r]; cW1[r] = 2.0 * d_c * W1[r];
                bW0[r] = d_b * W0[r]; bW1[r] = d_b * W1[r];
            }
            for (int r = 0; r < 3; r++)
                for (int cc = r; cc < 3; cc++)
                    H[3 * r + cc] += aW0[r] * W0[cc] + bW0[r] * W1[cc] + bW1[r] * W0[cc] + cW1[r] * W1[cc];
        }
        // ---- colour -> SH coefficients and view direction -> mean (py_compute_color).  The direction
        // is normalised in fp64; the SH polynomial, its gradient and the 48-term sums are fp32 (an
        // outer product and short dot products: 1e-6 relative is far inside the 1e-4 bar, at half
        // the registers and issue cycles of fp64).
        double vx = x - cam.cam_pos[0], vy = y - cam.cam_pos[1], vz = z - cam.cam_pos[2];
        const double vn_inv = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
        const double ddx_n = vx * vn_inv, ddy_n = vy * vn_inv, ddz_n = vz * vn_inv;
        const float dx = (float)ddx_n, dy = (float)ddy_n, dz = (float)ddz_n;
        float basis[16], bx[16], by[16], bz[16];
#pragma unroll
        for (int k = 0; k < 16; k++) basis[k] = bx[k] = by[k] = bz[k] = 0.0f;
        constexpr float C0 = (float)SH_C0, C1 = (float)SH_C1;
        constexpr float C20 = (float)SH_C2_0, C21 = (float)SH_C2_1, C22 = (float)SH_C2_2, C23 = (float)SH_C2_3,
                        C24 = (float)SH_C2_4;
        constexpr float C30 = (float)SH_C3_0, C31 = (float)SH_C3_1, C32 = (float)SH_C3_2, C33 = (float)SH_C3_3,
                        C34 = (float)SH_C3_4, C35 = (float)SH_C3_5, C36 = (float)SH_C3_6;
        basis[0] = C0;
        if (degree > 0) {
            basis[1] = -dy * C1; by[1] = -C1;
            basis[2] = dz * C1;  bz[2] = C1;
            basis[3] = -dx * C1; bx[3] = -C1;
            if (degree > 1) {
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                basis[4] = C20 * xy; bx[4] = C20 * dy; by[4] = C20 * dx;
                basis[5] = C21 * yz; by[5] = C21 * dz; bz[5] = C21 * dy;
                basis[6] = C22 * (2.0f * zz - xx - yy);
                bx[6] = -2.0f * C22 * dx; by[6] = -2.0f * C22 * dy; bz[6] = 4.0f * C22 * dz;
                basis[7] = C23 * xz; bx[7] = C23 * dz; bz[7] = C23 * dx;
                basis[8] = C24 * (xx - yy); bx[8] = 2.0f * C24 * dx; by[8] = -2.0f * C24 * dy;
                if (degree > 2) {
                    basis[9] = C30 * dy * (3.0f * xx - yy);
                    bx[9] = C30 * 6.0f * xy; by[9] = C30 * (3.0f * xx - 3.0f * yy);
                    basis[10] = C31 * xy * dz;
                    bx[10] = C31 * yz; by[10] = C31 * xz; bz[10] = C31 * xy;
                    basis[11] = C32 * dy * (4.0f * zz - xx - yy);
                    bx[11] = C32 * (-2.0f * xy); by[11] = C32 * (4.0f * zz - xx - 3.0f * yy); bz[11] = C32 * 8.0f * yz;
                    basis[12] = C33 * dz * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                    bx[12] = C33 * (-6.0f * xz); by[12] = C33 * (-6.0f * yz); bz[12] = C33 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                    basis[13] = C34 * dx * (4.0f * zz - xx - yy);
                    bx[13] = C34 * (4.0f * zz - 3.0f * xx - yy); by[13] = C34 * (-2.0f * xy); bz[13] = C34 * 8.0f * xz;
                    basis[14] = C35 * dz * (xx - yy);
                    bx[14] = C35 * 2.0f * xz; by[14] = C35 * (-2.0f * yz); bz[14] = C35 * (xx - yy);
                    basis[15] = C36 * dx * (xx - 3.0f * yy);
                    bx[15] = C36 * (3.0f * xx - 3.0f * yy); by[15] = C36 * (-6.0f * xy);
                }
            }
        }
        const int nb = (degree + 1) * (degree + 1);
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;   // dL/d(dir)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float dcol = in.clamp[ch] ? 0.0f : in_col[ch];
            float ex = 0.f, ey = 0.f, ez = 0.f;
#pragma unroll
            for (int vv = 0; vv < 4; vv++) {
                const float4 f = s_sh4[ch * 4 + vv][threadIdx.x];
                const float coef[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int k = 4 * vv + e;
                    if (k < nb) {
                        ex = __builtin_fmaf(bx[k], coef[e], ex);
                        ey = __builtin_fmaf(by[k], coef[e], ey);
                        ez = __builtin_fmaf(bz[k], coef[e], ez);
                        acc_sh[ch * 16 + k] = __builtin_fmaf(dcol, basis[k], acc_sh[ch * 16 + k]);
                    }
                }
            }
            ddx = __builtin_fmaf(dcol, ex, ddx);
            ddy = __builtin_fmaf(dcol, ey, ddy);
            ddz = __builtin_fmaf(dcol, ez, ddz);
        }
        // dir = v/|v|: dL/dv = (dL/ddir - dir (dir . dL/ddir)) / |v|   (dnormvdv, setup_gpu.py:256-266)
        const double dot = ddx_n * ddx + ddy_n * ddy + ddz_n * ddz;
        g_mean[0] += ((double)ddx - ddx_n * dot) * vn_inv;
        g_mean[1] += ((double)ddy - ddy_n * dot) * vn_inv;
        g_mean[2] += ((double)ddz - ddz_n * dot) * vn_inv;
    }

    if (any) {
        // ---- Sigma -> M: dM = (G + G^T) M
        H[3] = H[1]; H[6] = H[2]; H[7] = H[5];      // the loop keeps the upper triangle
        double dM[9];
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++)
                dM[3 * r + cc] = H[3 * r] * M[cc] + H[3 * r + 1] * M[3 + cc] + H[3 * r + 2] * M[6 + cc];

        // ---- M = R S': d(log scale_k) = (R^T dM)_kk * scale_mod * s_k
        for (int k = 0; k < 3; k++) {
            const double dS_kk = R[k] * dM[k] + R[3 + k] * dM[3 + k] + R[6 + k] * dM[6 + k];
            g_ls[k] = dS_kk * sm * sc[k];
        }
        // dR = dM S'^T
        double dR[9];
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++)
                dR[3 * r + cc] = dM[3 * r] * S[3 * cc] + dM[3 * r + 1] * S[3 * cc + 1] + dM[3 * r + 2] * S[3 * cc + 2];
        // R = I + (2/n2) Q(q)  (normalise-then-divide-by-squared-norm is scale invariant)
        {
            double QD = 0.0;
            for (int k = 0; k < 9; k++) QD += Q[k] * dR[k];
            const double D00 = dR[0], D01 = dR[1], D02 = dR[2], D10 = dR[3], D11 = dR[4], D12 = dR[5],
                         D20 = dR[6], D21 = dR[7], D22 = dR[8];
            const double pw = -qz * D01 + qy * D02 + qz * D10 - qx * D12 - qy * D20 + qx * D21;
            const double px = qy * D01 + qz * D02 + qy * D10 - 2 * qx * D11 - qw * D12 + qz * D20 + qw * D21 - 2 * qx * D22;
            const double py = -2 * qy * D00 + qx * D01 + qw * D02 + qx * D10 + qz * D12 - qw * D20 + qz * D21 - 2 * qy * D22;
            const double pz = -2 * qz * D00 - qw * D01 + qx * D02 + qw * D10 - 2 * qz * D11 + qy * D12 + qx * D20 + qy * D21;
            const double k2 = 2.0 * s * QD / n2;   // (4 QD / n2^2)
            g_q[0] = s * pw - k2 * qw;
            g_q[1] = s * px - k2 * qx;
            g_q[2] = s * py - k2 * qy;
            g_q[3] = s * pz - k2 * qz;
        }
        if (ROWS && opacity && opacity_is_logit) {     // opacity = sigmoid(parameter)
            const float o = 1.0f / (1.0f + __expf(-opacity[i]));
            g_op *= o * (1.0f - o);
        }
    }

#ifdef GS_K8_NOSTORE
    if (g_op != 123456.0f) return;         // experiment: everything but the stores
#endif
    // The SH gradients of a full wave (64 x 192 B, contiguous) leave through the wave's columns of s_sh4, whose
    // coefficients are no longer needed: twelve stores of 1 KiB of whole lines instead of twelve stores of 64 quarter
    // lines 192 B apart (one wave's LDS operations execute in order: no barrier).
    const int lane = threadIdx.x & 63;
    const bool full_wave = i - lane + 64 <= N;
    const bool wave_any = __ballot(any) != 0;
    if (full_wave && (wave_any || !accumulate)) {
#pragma unroll
        for (int vv = 0; vv < 12; vv++)
            s_sh4[vv][threadIdx.x] = make_float4(acc_sh[4 * vv], acc_sh[4 * vv + 1], acc_sh[4 * vv + 2], acc_sh[4 * vv + 3]);
        float4 *dst = reinterpret_cast<float4 *>(dL_dsh + (size_t)(i - lane) * 48);
        const int col0 = threadIdx.x - lane;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int j = k * 64 + lane;                 // float4 j of the wave's 768: gaussian j / 12, quarter j % 12
            float4 w = s_sh4[j % 12][col0 + j / 12];
            if (accumulate) {
                const float4 old = dst[j];
                w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w;
            }
            dst[j] = w;
        }
    }
    if (accumulate) {
        if (any) {
            for (int k = 0; k < 3; k++) { dL_dmeans[3 * i + k] += (float)g_mean[k]; dL_dlog_scales[3 * i + k] += (float)g_ls[k]; }
            for (int k = 0; k < 4; k++) dL_dquats[4 * i + k] += (float)g_q[k];
            if (ROWS && dL_dopacity) dL_dopacity[i] += g_op;
            if (!full_wave) {
#pragma unroll
                for (int vv = 0; vv < 12; vv++) {
                    float4 w = out_sh[vv];
                    w.x += acc_sh[4 * vv]; w.y += acc_sh[4 * vv + 1]; w.z += acc_sh[4 * vv + 2]; w.w += acc_sh[4 * vv + 3];
                    out_sh[vv] = w;
                }
            }
        }
    } else {
        for (int k = 0; k < 3; k++) { dL_dmeans[3 * i + k] = (float)g_mean[k]; dL_dlog_scales[3 * i + k] = (float)g_ls[k]; }
        for (int k = 0; k < 4; k++) dL_dquats[4 * i + k] = (float)g_q[k];
        if (ROWS && dL_dopacity) dL_dopacity[i] = g_op;
        if (!full_wave) {
#pragma unroll
            for (int vv = 0; vv < 12; vv++)
                out_sh[vv] = make_float4(acc_sh[4 * vv], acc_sh[4 * vv + 1], acc_sh[4 * vv + 2], acc_sh[4 * vv + 3]);
        }
    }
}

bool bwd_fast(const BatchArgs &a) {
    if (getenv("GS_K8_GENERAL")) return false;      // tests compare the two forms
    for (int v = 0; v < a.n_views; v++)
        if (a.v[v].cam.degree != 3 || a.v[v].cam.apply_view) return false;
    return true;
}

int check_common(gs_ctx *ctx, int32_t N, const gs_camera *cam) {
    GS_REQUIRE(ctx && cam, "gs_backward_params: ctx/cam is NULL");
    GS_REQUIRE(N >= 0, "gs_backward_params: N=%d < 0", N);
    GS_REQUIRE(cam->sh_degree >= 0 && cam->sh_degree <= 3, "gs_backward_params: sh_degree %d not in 0..3",
               cam->sh_degree);
    GS_REQUIRE(cam->res_w > 0 && cam->res_h > 0 && cam->cam_width > 0 && cam->cam_height > 0 && cam->fx > 0 &&
                   cam->fy > 0, "gs_backward_params: invalid camera/frame description");
    return GS_OK;
}

}  // namespace

extern "C" int gs_backward_params(gs_ctx *ctx, int32_t N, const float *means, const float *log_scales,
                                  const float *quats, const float *sh, const gs_camera *cam,
                                  const int32_t *radii, const uint8_t *clamped, const float *dL_dcenters,
                                  const float *dL_dconics, const float *dL_dcolors, int32_t accumulate,
                                  float *dL_dmeans, float *dL_dlog_scales, float *dL_dquats, float *dL_dsh) {
    int rc = check_common(ctx, N, cam);
    if (rc) return rc;
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && log_scales && quats && sh && radii && clamped && dL_dcenters && dL_dconics &&
                   dL_dcolors && dL_dmeans && dL_dlog_scales && dL_dquats && dL_dsh,
               "gs_backward_params: NULL array argument");
    GS_REQUIRE(((uintptr_t)sh & 15) == 0 && ((uintptr_t)dL_dsh & 15) == 0,
               "gs_backward_params: sh and dL_dsh must be 16-byte aligned");
    BatchArgs args;
    memset(&args, 0, sizeof args);
    args.n_views = 1;
    gs_make_precam(cam, &args.v[0].cam);
    args.v[0].radii = radii; args.v[0].clamped = clamped;
    args.v[0].d_centers = dL_dcenters; args.v[0].d_conics = dL_dconics; args.v[0].d_colors = dL_dcolors;
    GS_HIP(hipSetDevice(ctx->device));
    { gs_prof_scope prof_(ctx, GS_K_PARAM_BWD);
      auto kern = bwd_fast(args) ? param_bwd_kernel<false, true> : param_bwd_kernel<false, false>;
      hipLaunchKernelGGL(kern, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, means,
                       log_scales, quats, sh, (const float *)nullptr, 0, args, accumulate, dL_dmeans, dL_dlog_scales,
                       dL_dquats, (float *)nullptr, dL_dsh); }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_backward_params_batch(gs_ctx *ctx, int32_t N, const float *means, const float *log_scales,
                                        const float *quats, const float *sh, const float *opacity,
                                        int32_t opacity_is_logit, int32_t n_views, const gs_view_backward *views,
                                        int32_t accumulate, float *dL_dmeans, float *dL_dlog_scales,
                                        float *dL_dquats, float *dL_dopacity, float *dL_dsh) {
    GS_REQUIRE(ctx, "gs_backward_params_batch: ctx is NULL");
    GS_REQUIRE(N >= 0 && n_views >= 0, "gs_backward_params_batch: negative N or n_views");
    GS_REQUIRE(n_views == 0 || views, "gs_backward_params_batch: views is NULL");
    if (N == 0) return GS_OK;
    GS_REQUIRE(means && log_scales && quats && sh && dL_dmeans && dL_dlog_scales && dL_dquats && dL_dsh,
               "gs_backward_params_batch: NULL array argument");
    GS_REQUIRE(((uintptr_t)sh & 15) == 0 && ((uintptr_t)dL_dsh & 15) == 0,
               "gs_backward_params_batch: sh and dL_dsh must be 16-byte aligned");
    for (int v = 0; v < n_views; v++) {
        int rc = check_common(ctx, N, views[v].cam);
        if (rc) return rc;
        GS_REQUIRE(views[v].radii && views[v].clamped && views[v].grad_rows,
                   "gs_backward_params_batch: view %d has a NULL array", v);
        GS_REQUIRE(((uintptr_t)views[v].grad_rows & 15) == 0, "gs_backward_params_batch: grad_rows must be 16-byte aligned");
        GS_REQUIRE(views[v].cam->res_h == views[0].cam->res_h,
                   "gs_backward_params_batch: all views of a batch must share result_size (scale_mod)");
        GS_REQUIRE(views[v].cam->flags == views[0].cam->flags, "gs_backward_params_batch: all views of a batch must share camera flags");
    }
    GS_HIP(hipSetDevice(ctx->device));
    // chunks of MAX_BATCH_VIEWS views; the first chunk writes (unless accumulate), the rest add
    int done = 0;
    do {
        BatchArgs args;
        memset(&args, 0, sizeof args);
        const int n = n_views - done < MAX_BATCH_VIEWS ? n_views - done : MAX_BATCH_VIEWS;
        args.n_views = n;
        for (int v = 0; v < n; v++) {
            const gs_view_backward &src = views[done + v];
            gs_make_precam(src.cam, &args.v[v].cam);
            args.v[v].radii = src.radii; args.v[v].clamped = src.clamped; args.v[v].rows = src.grad_rows;
        }
        { gs_prof_scope prof_(ctx, GS_K_PARAM_BWD);
          auto kern = bwd_fast(args) ? param_bwd_kernel<true, true> : param_bwd_kernel<true, false>;
          hipLaunchKernelGGL(kern, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, means,
                           log_scales, quats, sh, opacity, opacity_is_logit, args, (accumulate || done > 0) ? 1 : 0,
                           dL_dmeans, dL_dlog_scales, dL_dquats, dL_dopacity, dL_dsh); }
        GS_HIP(hipGetLastError());
        done += n;
    } while (done < n_views);
    return GS_OK;
}
