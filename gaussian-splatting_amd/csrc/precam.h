// Per-frame constants derived on the host with the reference's own formulas (functions.py:11-39,
// forward_pass.py:43-50), shared by K1 (preprocess.hip) and K8 (param_bwd.hip).
#pragma once
#include "common.h"

struct PreCam {
    double p00, p11;       // proj_mat[0,0], [1,1]  (functions.py:16-17)
    double Rt[9];          // camera_r^T, row-major (view_mat[:3,:3], functions.py:6)
    double t[3];
    double cam_pos[3];     // camera_r^T @ camera_t (forward_pass.py:47)
    double fx, fy;
    double scale_mod;      // 3 / result_size[1]    (forward_pass.py:50)
    int W, H, tile, degree;
    int apply_view, diag_scale;   // gs_camera.flags (0, 0 = strict reference)
    double inv_tile;       // 1 / tile; v * inv_tile == v / tile bit for bit when tile is a power of two (tile_pow2)
    int tile_pow2;
};

// forward_pass.py:8-21
constexpr double SH_C0 = 0.28209479177387814;
constexpr double SH_C1 = 0.4886025119029199;
constexpr double SH_C2_0 = 1.0925484305920792, SH_C2_1 = -1.0925484305920792,
                 SH_C2_2 = 0.31539156525252005, SH_C2_3 = -1.0925484305920792,
                 SH_C2_4 = 0.5462742152960396;
constexpr double SH_C3_0 = -0.5900435899266435, SH_C3_1 = 2.890611442640554,
                 SH_C3_2 = -0.4570457994644658, SH_C3_3 = 0.3731763325901154,
                 SH_C3_4 = -0.4570457994644658, SH_C3_5 = 1.445305721320277,
                 SH_C3_6 = -0.5900435899266435;

int gs_make_precam(const gs_camera *cam, PreCam *out);
