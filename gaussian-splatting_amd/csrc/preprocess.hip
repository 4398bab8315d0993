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
#include <stdlib.h>

namespace {

// int(min(size, max(0, v)) / tile) with Python's min/max semantics (forward_pass.py:116-119)
// FAST: the tile size is a power of two, so the division is the multiplication by its (exact) reciprocal
template <bool FAST>
__device__ inline int clamp_to_tile(double v, double size, double tile, double inv_tile) {
    double m = (v > 0.0) ? v : 0.0;
    m = (m < size) ? m : size;
    return (int)(FAST ? m * inv_tile : m / tile);
}

// everything of one gaussian that depends on the view; `coef` holds its 48 SH coefficients
struct PreOut {
    float cx, cy, col[3], con[3];
    uint8_t cl[3];
    int depth, rad, rect[4];
    double eW[6], eM[9], eT[3], eC2[3], eC3[6], eD[3];
};

// :74-76  the view-independent part of a gaussian: M = R(q) S and Sigma = M M^T.  Only scale_mod (3 / result height)
// and the diag_scale switch come from the frame, and every view of a batch shares them, so the batched kernel
// evaluates this ONCE per gaussian (three fp64 exps, a square root, four divisions and 54 multiply-adds less per
// view); same operations in the same order as inside the per-view chain, so the outputs do not change by a bit.
__device__ __forceinline__ void gaussian_cov(double sm, int diag_scale, double q0, double q1, double q2, double q3,
                                             const float (&ls)[3], double (&M)[9], double (&Sg)[9]) {
    // :74  rotation from the normalised quaternion (w,x,y,z); as_rotation_matrix divides by
    // the squared norm once more
    const double qn = sqrt(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
    const double qw = q0 / qn, qx = q1 / qn, qy = q2 / qn, qz = q3 / qn;
    const double s = 2.0 / (((qw * qw + qx * qx) + qy * qy) + qz * qz);
    const double R[9] = {1 - s * (qy * qy + qz * qz), s * (qx * qy - qz * qw), s * (qx * qz + qy * qw),
                         s * (qx * qy + qz * qw), 1 - s * (qx * qx + qz * qz), s * (qy * qz - qx * qw),
                         s * (qx * qz - qy * qw), s * (qy * qz + qx * qw), 1 - s * (qx * qx + qy * qy)};
    // :75  M = R @ ([[s0,1,1],[1,s1,1],[1,1,s2]] * scale_mod) (quirk Q4)
    const double so = diag_scale ? 0.0 : sm;       // off-diagonals: 1 * scale_mod in the reference
    const double S[9] = {exp((double)ls[0]) * sm, so, so,
                         so, exp((double)ls[1]) * sm, so,
                         so, so, exp((double)ls[2]) * sm};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            M[3 * r + c] = (R[3 * r] * S[c] + R[3 * r + 1] * S[3 + c]) + R[3 * r + 2] * S[6 + c];
    // :76  Sigma = M M^T
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            Sg[3 * r + c] = (M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1]) + M[3 * r + 2] * M[3 * c + 2];
}

// FAST (chosen per launch on the host, pre_fast()): SH degree 3, clip-space centre from world coordinates (the strict
// reference) and a power-of-two tile size for every view of the launch -- the degree and flag tests below fold away
// (they cost two selects per fp64 term of the colour sums otherwise) and the four tile divisions become exact
// multiplications.  Same operations on the same values: outputs identical to the general form bit for bit.
template <bool EXTRAS, bool FAST>
__device__ __forceinline__ void preprocess_one(const PreCam &cam, double x, double y, double z, const double (&M)[9],
                                               const double (&Sg)[9], const float (&coef)[48], PreOut &o) {
    // outputs of a culled gaussian are all zero (the reference pre-zeroes, forward_pass.py:29-41)
    o.cx = o.cy = 0.f;
    for (int k = 0; k < 3; k++) { o.col[k] = 0.f; o.con[k] = 0.f; o.cl[k] = 0; }
    o.depth = 0; o.rad = 0;
    for (int k = 0; k < 4; k++) o.rect[k] = 0;
    if (EXTRAS) {
        for (int k = 0; k < 6; k++) { o.eW[k] = 0; o.eC3[k] = 0; }
        for (int k = 0; k < 9; k++) o.eM[k] = 0;
        for (int k = 0; k < 3; k++) { o.eT[k] = 0; o.eC2[k] = 0; o.eD[k] = 0; }
    }

    do {
        // :64  view-space position
        const double tx = ((cam.Rt[0] * x + cam.Rt[1] * y) + cam.Rt[2] * z) + cam.t[0];
        const double ty = ((cam.Rt[3] * x + cam.Rt[4] * y) + cam.Rt[5] * z) + cam.t[1];
        const double tz = ((cam.Rt[6] * x + cam.Rt[7] * y) + cam.Rt[8] * z) + cam.t[2];

        // :59-61  clip-space centre from WORLD coordinates (quirk Q1); GS_CAM_APPLY_VIEW: from the view-space
        // position (the line the reference has commented out, :57)
        const bool view_space = !FAST && cam.apply_view;
        const int degree = FAST ? 3 : cam.degree;
        const double hx = view_space ? tx : x, hy = view_space ? ty : y, hz = view_space ? tz : z;
        const double mod_w = 1.0 / (hz + .00000001);
        const double ndc_x = (cam.p00 * hx) * mod_w;
        const double ndc_y = (cam.p11 * hy) * mod_w;

        // :65-67  near cull at z <= 2 (quirk Q6)
        if (tz <= 2.0 || isnan(tz)) break;

        // :70-71  depth key = int(z*1000) wrapped to int32 (quirk Q7)
        {
            const double d = trunc(tz * 1000.0);
            if (d >= -2147483648.0 && d < 2147483648.0) {
                o.depth = (int)d;
            } else if (isfinite(d)) {
                const double w = fmod(d + 2147483648.0, 4294967296.0);
                o.depth = (int)((w < 0 ? w + 4294967296.0 : w) - 2147483648.0);
            }
        }

        // :78-82  W = J^T @ camera_r^T; only rows 0,1 reach cov_2d[0:2,0:2] (quirk Q5)
        const double j00 = cam.fx / tz;
        double j11 = j00;                               // square pixels: the same quotient
        if (cam.fy != cam.fx) j11 = cam.fy / tz;
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
        if (EXTRAS) {
            for (int k = 0; k < 3; k++) { o.eW[k] = W0[k]; o.eW[3 + k] = W1[k]; }
            for (int k = 0; k < 9; k++) o.eM[k] = M[k];
            o.eT[0] = tx; o.eT[1] = ty; o.eT[2] = tz;
            o.eC2[0] = a; o.eC2[1] = b; o.eC2[2] = c;
            o.eC3[0] = Sg[0]; o.eC3[1] = Sg[1]; o.eC3[2] = Sg[2]; o.eC3[3] = Sg[4]; o.eC3[4] = Sg[5]; o.eC3[5] = Sg[8];
        }

        // :97-103  conic
        const double det = a * c - b * b;
        if (det == 0.0) break;
        const double det_inv = 1.0 / det;
        o.con[0] = (float)(c * det_inv);
        o.con[1] = (float)(-b * det_inv);
        o.con[2] = (float)(a * det_inv);

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
        o.rect[0] = clamp_to_tile<FAST>(cx - radius, cam.W, cam.tile, cam.inv_tile);
        o.rect[1] = clamp_to_tile<FAST>(cy - radius, cam.H, cam.tile, cam.inv_tile);
        o.rect[2] = clamp_to_tile<FAST>(cx + radius + cam.tile - 1, cam.W, cam.tile, cam.inv_tile);
        o.rect[3] = clamp_to_tile<FAST>(cy + radius + cam.tile - 1, cam.H, cam.tile, cam.inv_tile);

        // :122-123  off-screen cull
        if (cx + radius < 0 || cx - radius > cam.W || cy + radius < 0 || cy - radius > cam.H) break;

        o.rad = (int)radius;
        o.cx = (float)cx;
        o.cy = (float)cy;

        // :127 -> py_compute_color (:168-215)
        double dx = x - cam.cam_pos[0], dy = y - cam.cam_pos[1], dz = z - cam.cam_pos[2];
        // The colour is a float output (tolerance, not bits; the reference's own sums associate differently, :178-203):
        // one division and three products for the direction, fused multiply-adds for the 48-term sums.
        const double dn_inv = 1.0 / sqrt((dx * dx + dy * dy) + dz * dz);
        dx *= dn_inv; dy *= dn_inv; dz *= dn_inv;
        if (EXTRAS) { o.eD[0] = dx; o.eD[1] = dy; o.eD[2] = dz; }
        double basis[16];
        basis[0] = SH_C0;
        if (degree > 0) {
            basis[1] = -dy * SH_C1; basis[2] = dz * SH_C1; basis[3] = -dx * SH_C1;
            if (degree > 1) {
                const double xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                basis[4] = SH_C2_0 * xy; basis[5] = SH_C2_1 * yz;
                basis[6] = SH_C2_2 * (2.0 * zz - xx - yy);
                basis[7] = SH_C2_3 * xz; basis[8] = SH_C2_4 * (xx - yy);
                if (degree > 2) {
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
        const int nb = (degree + 1) * (degree + 1);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) {
                    float cf = coef[16 * ch + k];
#ifdef GS_K1_CVT_IN_LOOP
                    asm volatile("" : "+v"(cf));        // keep the coefficient a float across views (48 VGPRs, not 96)
#endif
                    r = __builtin_fma(basis[k], (double)cf, r);
                }
            r += .5;
            if (r < 0) { r = 0; o.cl[ch] = 1; }
            o.col[ch] = (float)r;
        }
    } while (0);
}

__device__ __forceinline__ void load_params(int i, const float *__restrict__ means, const float *__restrict__ log_scales,
                                            const float *__restrict__ quats, const float *__restrict__ sh, double &x,
                                            double &y, double &z, double (&q)[4], float (&ls)[3], float (&coef)[48]) {
    x = means[3 * i]; y = means[3 * i + 1]; z = means[3 * i + 2];
    for (int k = 0; k < 4; k++) q[k] = quats[4 * i + k];
    for (int k = 0; k < 3; k++) ls[k] = log_scales[3 * i + k];
    const float4 *sh4 = reinterpret_cast<const float4 *>(sh + (size_t)i * 48);
#pragma unroll
    for (int v = 0; v < 12; v++) {
        const float4 f = sh4[v];
        coef[4 * v] = f.x; coef[4 * v + 1] = f.y; coef[4 * v + 2] = f.z; coef[4 * v + 3] = f.w;
    }
}

struct ViewOut {
    float *centers, *colors, *conics;
    int32_t *depths, *tiles_touched, *radii;
    uint8_t *clamped;
    float4 *records;     // record-writing form: replaces centers / colors / conics
};

__device__ __forceinline__ void store_view(int i, const ViewOut &w, const PreOut &o) {
    w.centers[2 * i] = o.cx; w.centers[2 * i + 1] = o.cy;
    w.depths[i] = o.depth;
    w.radii[i] = o.rad;
    reinterpret_cast<int4 *>(w.tiles_touched)[i] = make_int4(o.rect[0], o.rect[1], o.rect[2], o.rect[3]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        w.colors[3 * i + k] = o.col[k];
        w.conics[3 * i + k] = o.con[k];
        w.clamped[3 * i + k] = o.cl[k];
    }
}

__global__ __launch_bounds__(256) void preprocess_kernel(
    int N, const float *__restrict__ means, const float *__restrict__ log_scales,
    const float *__restrict__ quats, const float *__restrict__ sh, PreCam cam, ViewOut out,
    float *__restrict__ Ws, float *__restrict__ Ms, float *__restrict__ Ts, float *__restrict__ cov2d,
    float *__restrict__ cov3d, float *__restrict__ dirs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double x, y, z, q[4];
    float ls[3], coef[48];
    load_params(i, means, log_scales, quats, sh, x, y, z, q, ls, coef);
    PreOut o;
    double M[9], Sg[9];
    gaussian_cov(cam.scale_mod, cam.diag_scale, q[0], q[1], q[2], q[3], ls, M, Sg);
    preprocess_one<true, false>(cam, x, y, z, M, Sg, coef, o);
    store_view(i, out, o);
    if (Ws) for (int k = 0; k < 6; k++) Ws[6 * i + k] = (float)o.eW[k];
    if (Ms) for (int k = 0; k < 9; k++) Ms[9 * i + k] = (float)o.eM[k];
    if (Ts) for (int k = 0; k < 3; k++) Ts[3 * i + k] = (float)o.eT[k];
    if (cov2d) for (int k = 0; k < 3; k++) cov2d[3 * i + k] = (float)o.eC2[k];
    if (cov3d) for (int k = 0; k < 6; k++) cov3d[6 * i + k] = (float)o.eC3[k];
    if (dirs) for (int k = 0; k < 3; k++) dirs[3 * i + k] = (float)o.eD[k];
}

// Multi-view form (no counterpart in the reference, which renders one view per iteration,
// train.py:18-28): the 236 bytes of parameters are read once and every view of the batch is
// projected from registers.
constexpr int MAX_PRE_VIEWS = 8;
struct PreBatch {
    PreCam cam[MAX_PRE_VIEWS];
    ViewOut out[MAX_PRE_VIEWS];
    int n_views;
};

// RECORDS: instead of centers / colors / conics each view gets the 64-byte blend records (common.h) the
// blend and culled-binning kernels read, built from the activated opacities: no pack pass per view.
#ifdef GS_K1_WAVES
#define GS_K1_OCC __attribute__((amdgpu_waves_per_eu(GS_K1_WAVES, GS_K1_WAVES)))
#else
#define GS_K1_OCC
#endif
template <bool RECORDS, bool FAST>
__global__ __launch_bounds__(256) GS_K1_OCC void preprocess_batch_kernel(int N, const float *__restrict__ means,
                                                               const float *__restrict__ log_scales,
                                                               const float *__restrict__ quats,
                                                               const float *__restrict__ sh,
                                                               const float *__restrict__ opacities, const PreBatch b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double x, y, z, q[4];
    float ls[3], coef[48];
    // (the SH coefficients as twelve coalesced 1 KiB loads through wave-private LDS instead of twelve loads of 64
    // quarter lines: 0.190 against 0.191 ms per 8 views -- the loads are not what the kernel waits for)
    load_params(i, means, log_scales, quats, sh, x, y, z, q, ls, coef);
    const float opac = RECORDS ? opacities[i] : 0.0f;
    // M and Sigma once per gaussian: the host has checked that the views share scale_mod and the scale switch
    double M[9], Sg[9];
    gaussian_cov(b.cam[0].scale_mod, b.cam[0].diag_scale, q[0], q[1], q[2], q[3], ls, M, Sg);
#pragma unroll 1
    for (int v = 0; v < b.n_views; v++) {
        PreOut o;
        preprocess_one<false, FAST>(b.cam[v], x, y, z, M, Sg, coef, o);
#ifdef GS_K1_NOSTORE
        if (o.depth != 0x7ffffff1) continue;      // experiment: the arithmetic without the stores
#endif
        if (RECORDS) {
            const ViewOut &w = b.out[v];
            w.depths[i] = o.depth;
            w.radii[i] = o.rad;
            reinterpret_cast<int4 *>(w.tiles_touched)[i] = make_int4(o.rect[0], o.rect[1], o.rect[2], o.rect[3]);
#pragma unroll
            for (int k = 0; k < 3; k++) w.clamped[3 * i + k] = o.cl[k];
            float4 geo, con, rgb;
            gs_make_record(o.cx, o.cy, opac, o.con[0], o.con[1], o.con[2], o.col[0], o.col[1], o.col[2], geo, con, rgb);
            // The wave's 64 records go through wave-private LDS ([quarter][lane], pitch 68 float4: conflict-free both
            // ways; one wave's LDS operations execute in order, so no barrier) and leave as four stores of 1 KiB of
            // whole lines each, instead of four stores of 64 quarter lines: 0.228 -> 0.202 ms per 8 views.  (Leaving
            // the unused fourth quarter unwritten costs a read-modify-write per line: 0.303 ms.)
            __shared__ float4 s_rec[4][4][68];
            const int lane = threadIdx.x & 63;
            float4 (*st)[68] = s_rec[threadIdx.x >> 6];
            // fourth quarter: the tile rect again, as bits (the culled tile count of large scenes gathers the record's
            // line anyway and finds the rect in it instead of in a second gathered line)
            const float4 rq = make_float4(__int_as_float(o.rect[0]), __int_as_float(o.rect[1]), __int_as_float(o.rect[2]),
                                            __int_as_float(o.rect[3]));
            if (i - lane + 64 <= N) {
                st[0][lane] = geo; st[1][lane] = con; st[2][lane] = rgb; st[3][lane] = rq;
                float4 *dst = w.records + (size_t)(i - lane) * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = k * 64 + lane;
                    dst[j] = st[j & 3][j >> 2];
                }
            } else {                                   // the last, partial wave
                float4 *r = w.records + (size_t)i * 4;
                r[0] = geo; r[1] = con; r[2] = rgb; r[3] = rq;
            }
        } else {
            store_view(i, b.out[v], o);
        }
    }
}

// every view of the launch qualifies for the FAST form of preprocess_one
bool pre_fast(const PreBatch &b) {
    if (getenv("GS_K1_GENERAL")) return false;      // tests compare the two forms
    for (int v = 0; v < b.n_views; v++)
        if (b.cam[v].degree != 3 || b.cam[v].apply_view || !b.cam[v].tile_pow2) return false;
    return true;
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
    pc.apply_view = (cam->flags & GS_CAM_APPLY_VIEW) ? 1 : 0;
    pc.diag_scale = (cam->flags & GS_CAM_DIAG_SCALE) ? 1 : 0;
    pc.inv_tile = 1.0 / pc.tile;
    pc.tile_pow2 = (pc.tile & (pc.tile - 1)) == 0;
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
    ViewOut vo{centers, colors, conics, depths, tiles_touched, radii, clamped};
    { gs_prof_scope prof_(ctx, GS_K_PREPROCESS); hipLaunchKernelGGL(preprocess_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N,
                       means, log_scales, quats, sh, pc, vo, Ws, Ms, Ts, cov2d, cov3d, dirs); }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

extern "C" int gs_preprocess_batch(gs_ctx *ctx, int32_t N, const float *means, const float *log_scales,
                                   const float *quats, const float *sh, int32_t n_views,
                                   const gs_view_preprocess *views) {
    GS_REQUIRE(ctx, "gs_preprocess_batch: ctx is NULL");
    GS_REQUIRE(N >= 0 && n_views >= 0, "gs_preprocess_batch: negative N or n_views");
    GS_REQUIRE(n_views == 0 || views, "gs_preprocess_batch: views is NULL");
    if (N == 0 || n_views == 0) return GS_OK;
    GS_REQUIRE(means && log_scales && quats && sh, "gs_preprocess_batch: NULL parameter array");
    GS_REQUIRE(((uintptr_t)sh & 15) == 0, "gs_preprocess_batch: sh must be 16-byte aligned");
    for (int v = 0; v < n_views; v++) {
        const gs_view_preprocess &w = views[v];
        GS_REQUIRE(w.cam, "gs_preprocess_batch: view %d has no camera", v);
        GS_REQUIRE(w.cam->sh_degree >= 0 && w.cam->sh_degree <= 3, "gs_preprocess_batch: sh_degree %d not in 0..3",
                   w.cam->sh_degree);
        GS_REQUIRE(w.cam->res_w > 0 && w.cam->res_h > 0 && w.cam->tile_size > 0 && w.cam->cam_width > 0 &&
                       w.cam->cam_height > 0 && w.cam->fx > 0 && w.cam->fy > 0,
                   "gs_preprocess_batch: invalid camera/frame description in view %d", v);
        GS_REQUIRE(w.centers && w.depths && w.colors && w.conics && w.clamped && w.tiles_touched && w.radii,
                   "gs_preprocess_batch: view %d has a NULL output", v);
        GS_REQUIRE(((uintptr_t)w.tiles_touched & 15) == 0, "gs_preprocess_batch: tiles_touched must be 16-byte aligned");
    }
    GS_HIP(hipSetDevice(ctx->device));
    for (int done = 0; done < n_views; done += 0) {
        PreBatch b;
        b.n_views = 0;
        // a launch takes up to 8 consecutive views that share what Sigma depends on (the result height's scale_mod
        // and the scale switch): the kernel forms M and Sigma once per gaussian for all of them
        while (done + b.n_views < n_views && b.n_views < MAX_PRE_VIEWS) {
            const int v = b.n_views;
            const gs_view_preprocess &w = views[done + v];
            gs_make_precam(w.cam, &b.cam[v]);
            if (v > 0 && (b.cam[v].scale_mod != b.cam[0].scale_mod || b.cam[v].diag_scale != b.cam[0].diag_scale)) break;
            b.out[v] = ViewOut{w.centers, w.colors, w.conics, w.depths, w.tiles_touched, w.radii, w.clamped, nullptr};
            b.n_views++;
        }
        done += b.n_views;
        { gs_prof_scope prof_(ctx, GS_K_PREPROCESS);
          auto kern = pre_fast(b) ? preprocess_batch_kernel<false, true> : preprocess_batch_kernel<false, false>;
          hipLaunchKernelGGL(kern, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, means, log_scales, quats, sh,
                             (const float *)nullptr, b); }
        GS_HIP(hipGetLastError());
    }
    return GS_OK;
}

extern "C" int gs_preprocess_batch_records(gs_ctx *ctx, int32_t N, const float *means, const float *log_scales,
                                           const float *quats, const float *sh, const float *opacities,
                                           int32_t n_views, const gs_view_records *views) {
    GS_REQUIRE(ctx, "gs_preprocess_batch_records: ctx is NULL");
    GS_REQUIRE(N >= 0 && n_views >= 0, "gs_preprocess_batch_records: negative N or n_views");
    GS_REQUIRE(n_views == 0 || views, "gs_preprocess_batch_records: views is NULL");
    if (N == 0 || n_views == 0) return GS_OK;
    GS_REQUIRE(means && log_scales && quats && sh && opacities, "gs_preprocess_batch_records: NULL parameter array");
    GS_REQUIRE(((uintptr_t)sh & 15) == 0, "gs_preprocess_batch_records: sh must be 16-byte aligned");
    for (int v = 0; v < n_views; v++) {
        const gs_view_records &w = views[v];
        GS_REQUIRE(w.cam, "gs_preprocess_batch_records: view %d has no camera", v);
        GS_REQUIRE(w.cam->sh_degree >= 0 && w.cam->sh_degree <= 3, "gs_preprocess_batch_records: sh_degree %d not in 0..3",
                   w.cam->sh_degree);
        GS_REQUIRE(w.cam->res_w > 0 && w.cam->res_h > 0 && w.cam->tile_size > 0 && w.cam->cam_width > 0 &&
                       w.cam->cam_height > 0 && w.cam->fx > 0 && w.cam->fy > 0,
                   "gs_preprocess_batch_records: invalid camera/frame description in view %d", v);
        GS_REQUIRE(w.records && w.depths && w.clamped && w.tiles_touched && w.radii,
                   "gs_preprocess_batch_records: view %d has a NULL output", v);
        GS_REQUIRE(((uintptr_t)w.tiles_touched & 15) == 0 && ((uintptr_t)w.records & 15) == 0,
                   "gs_preprocess_batch_records: tiles_touched and records must be 16-byte aligned");
    }
    GS_HIP(hipSetDevice(ctx->device));
    for (int done = 0; done < n_views; done += 0) {
        PreBatch b;
        b.n_views = 0;
        while (done + b.n_views < n_views && b.n_views < MAX_PRE_VIEWS) {      // as in gs_preprocess_batch
            const int v = b.n_views;
            const gs_view_records &w = views[done + v];
            gs_make_precam(w.cam, &b.cam[v]);
            if (v > 0 && (b.cam[v].scale_mod != b.cam[0].scale_mod || b.cam[v].diag_scale != b.cam[0].diag_scale)) break;
            b.out[v] = ViewOut{nullptr, nullptr, nullptr, w.depths, w.tiles_touched, w.radii, w.clamped,
                               reinterpret_cast<float4 *>(w.records)};
            b.n_views++;
        }
        done += b.n_views;
        { gs_prof_scope prof_(ctx, GS_K_PREPROCESS);
          auto kern = pre_fast(b) ? preprocess_batch_kernel<true, true> : preprocess_batch_kernel<true, false>;
          hipLaunchKernelGGL(kern, dim3(gs_div_up(N, 256)), dim3(256), 0, ctx->stream, N, means, log_scales, quats, sh,
                             opacities, b); }
        GS_HIP(hipGetLastError());
    }
    return GS_OK;
}
