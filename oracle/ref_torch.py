"""ORACLE (test infrastructure only): differentiable fp64 torch restatement of the reference's
forward path, used to pin gradients by autograd.

The reference's own backward (setup_gpu.py:105-303, backward_pass.py:19-68) cannot run here and is
undefined at HEAD, so gradients are "parity unpinned by the
reference": the bar is d(forward)/d(parameters) of THIS restatement of the live forward path
(forward_pass.py:25-136, rasterize.c:82-131), evaluated by torch.autograd in float64, plus central
finite differences in tests.  The forward here is checked against the golden vectors first
(tests/test_oracle_golden.py), so what is differentiated is the reference's forward.

Discrete quantities (depth keys, radii, tile rects, per-tile order, skip/break decisions) carry no
gradient; alpha = min(.99, o*G) has zero gradient where the clamp is active.
"""
import math
import torch

from . import ref_np

SH_C0, SH_C1 = ref_np.SH_C0, ref_np.SH_C1
SH_C2 = [float(v) for v in ref_np.SH_C2]
SH_C3 = [float(v) for v in ref_np.SH_C3]


def quat_to_rotmat(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    s = 2.0 / (w * w + x * x + y * y + z * z)
    R = torch.stack([
        1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
        s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
        s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def sh_basis(d, degree):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    one = torch.ones_like(x)
    b = [SH_C0 * one]
    if degree > 0:
        b += [-y * SH_C1, z * SH_C1, -x * SH_C1]
        if degree > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            b += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * xz,
                  SH_C2[4] * (xx - yy)]
            if degree > 2:
                b += [SH_C3[0] * y * (3.0 * xx - yy), SH_C3[1] * xy * z,
                      SH_C3[2] * y * (4.0 * zz - xx - yy),
                      SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy),
                      SH_C3[4] * x * (4.0 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                      SH_C3[6] * x * (xx - 3.0 * yy)]
    while len(b) < 16:
        b.append(torch.zeros_like(x))
    return torch.stack(b, dim=1)


def preprocess(means, log_scales, quats, sh, degree, fx, fy, cam_width, cam_height,
               camera_r, camera_t, result_size, apply_view=False, diag_scale=False):
    """Differentiable part of forward_pass.py:56-127: centers[N,2] (pixels), conics[N,3],
    colors[N,3] for every gaussian (no culling -- the caller masks with the discrete outputs of
    ref_np.preprocess)."""
    W, H = int(result_size[0]), int(result_size[1])
    R_c = torch.as_tensor(camera_r, dtype=torch.float64)
    t_c = torch.as_tensor(camera_t, dtype=torch.float64)
    fov_x = ref_np.focal_to_fov(fx, cam_width)
    fov_y = ref_np.focal_to_fov(fx, cam_height)
    proj = torch.as_tensor(ref_np.get_projection_matrix(fov_x, fov_y))
    cam_pos = R_c.T @ t_c
    scale_mod = (1.0 / H) * 3
    N = means.shape[0]

    tview = means @ R_c + t_c                      # rows of (R^T p + t)
    # 2D centre straight from world coordinates (quirk Q1, forward_pass.py:58-61,114); apply_view: from tview
    hom = tview if apply_view else means
    mod_w = 1.0 / (hom[:, 2] * proj[3, 2] + .00000001)
    ndc_x = hom[:, 0] * proj[0, 0] * mod_w
    ndc_y = hom[:, 1] * proj[1, 1] * mod_w
    cx = ((ndc_x + 1.0) * W - 1.0) * .5
    cy = ((ndc_y + 1.0) * H - 1.0) * .5

    qn = quats / quats.norm(dim=1, keepdim=True)
    R = quat_to_rotmat(qn)
    s = torch.exp(log_scales)
    S = torch.diag_embed(s)
    if not diag_scale:
        S = S + torch.ones(N, 3, 3, dtype=torch.float64) - torch.diag_embed(torch.ones(N, 3, dtype=torch.float64))
    S = S * scale_mod
    M = R @ S
    cov3 = M @ M.transpose(1, 2)
    tz = tview[:, 2]
    # top-left 2x2 of W Sigma W^T only needs rows 0,1 of W = J^T R_c^T (quirk Q5)
    w0 = (fx / tz)[:, None] * R_c.T[0][None, :]
    w1 = (fy / tz)[:, None] * R_c.T[1][None, :]
    a = torch.einsum('ni,nij,nj->n', w0, cov3, w0) + .3
    b = torch.einsum('ni,nij,nj->n', w0, cov3, w1)
    c = torch.einsum('ni,nij,nj->n', w1, cov3, w1) + .3
    det = a * c - b * b
    det_inv = 1.0 / det
    conics = torch.stack([c * det_inv, -b * det_inv, a * det_inv], dim=1)

    dvec = means - cam_pos
    dirs = dvec / dvec.norm(dim=1, keepdim=True)
    col = torch.einsum('nk,nck->nc', sh_basis(dirs, degree), sh) + .5
    colors = torch.where(col < 0, torch.zeros_like(col), col)
    return torch.stack([cx, cy], dim=1), conics, colors


def blend(centers, colors, opacities, conics, vals, ranges, result_size, tile_size=16,
          background=(0, 0, 0), pixel_offset=0.5, forced=None):
    """Differentiable rasterize.c:82-131 over the whole frame, vectorised across tiles and the
    256 pixels of a tile; sequential over list position.  Returns image[W,H,3].
    forced = ref_np.device_decisions(...): the skip / break branches are not decided here but taken as the
    device took them (which entries a pixel visited, whether it left through the break, which visited pairs were
    valid), so the derivative is that of the very function the kernels evaluated -- on every pixel."""
    W, H = int(result_size[0]), int(result_size[1])
    tx_n, ty_n = ref_np.tile_grid(result_size, tile_size)
    tiles = tx_n * ty_n
    ranges_t = torch.as_tensor(ranges, dtype=torch.int64)
    vals_t = torch.as_tensor(vals, dtype=torch.int64)
    lens = ranges_t[:, 1] - ranges_t[:, 0]
    L = int(lens.max()) if tiles else 0
    bg = torch.as_tensor(background, dtype=torch.float64)

    tid = torch.arange(tiles)
    tx, ty = tid // ty_n, tid % ty_n
    lx = torch.arange(tile_size).repeat_interleave(tile_size)     # [256] local x (x-major)
    ly = torch.arange(tile_size).repeat(tile_size)
    px = (tx[:, None] * tile_size + lx[None, :]).to(torch.float64)   # [tiles,256]
    py = (ty[:, None] * tile_size + ly[None, :]).to(torch.float64)

    T = torch.ones(tiles, tile_size * tile_size, dtype=torch.float64)
    C = torch.zeros(tiles, tile_size * tile_size, 3, dtype=torch.float64)
    done = torch.zeros(tiles, tile_size * tile_size, dtype=torch.bool)
    for j in range(L):
        has = lens > j                                            # [tiles]
        idx = torch.where(has, ranges_t[:, 0] + j, torch.zeros_like(lens))
        g = vals_t[idx] if len(vals_t) else torch.zeros_like(idx)
        dx = centers[g, 0][:, None] - px + pixel_offset
        dy = centers[g, 1][:, None] - py + pixel_offset
        cn = conics[g]
        power = -.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
        # pairs with power > 0 are skipped below; clamping keeps exp() finite there so that autograd does not
        # turn their masked-out branch into 0 * inf = NaN (indefinite conics)
        alpha = torch.clamp(opacities[g][:, None] * torch.exp(torch.clamp(power, max=0.0)), max=.99)
        test_T = T * (1.0 - alpha)
        if forced is None:
            active = has[:, None] & ~done & ~(power > 0) & ~(alpha < (1.0 / 255))
            brk = active & (test_T < .0001)
        else:
            seen = torch.as_tensor(forced["visited"] > j)
            active = has[:, None] & seen & torch.as_tensor(forced["valid"][:, j, :])
            brk = active & torch.as_tensor(forced["broke"] & (forced["visited"] == j + 1))
        done = done | brk
        take = active & ~brk
        w = torch.where(take, alpha * T, torch.zeros_like(T))
        C = C + w[:, :, None] * colors[g][:, None, :]
        T = torch.where(take, test_T, T)
    C = C + T[:, :, None] * bg[None, None, :]
    # scatter tiles back to image[W,H,3]; drop pixels beyond the frame
    img = torch.zeros(tx_n * tile_size, ty_n * tile_size, 3, dtype=torch.float64)
    Cv = C.reshape(tx_n, ty_n, tile_size, tile_size, 3).permute(0, 2, 1, 3, 4)
    img = Cv.reshape(tx_n * tile_size, ty_n * tile_size, 3)
    return img[:W, :H]


def render(means, log_scales, quats, sh, opacities, degree, fx, fy, cam_width, cam_height,
           camera_r, camera_t, result_size, tile_size=16, background=(0, 0, 0), pixel_offset=0.5,
           apply_view=False, diag_scale=False):
    """Full differentiable forward: discrete decisions from ref_np.preprocess / bin_duplicates on
    the detached parameters, continuous values from the torch graph."""
    disc = ref_np.preprocess(means.detach().numpy(), log_scales.detach().numpy(),
                             quats.detach().numpy(), sh.detach().numpy(), degree, fx, fy,
                             cam_width, cam_height, camera_r, camera_t, result_size, tile_size,
                             apply_view=apply_view, diag_scale=diag_scale)
    keys, vals, ranges = ref_np.bin_duplicates(disc['tiles_touched'], disc['depths'], result_size,
                                               tile_size)
    centers, conics, colors = preprocess(means, log_scales, quats, sh, degree, fx, fy, cam_width,
                                         cam_height, camera_r, camera_t, result_size, apply_view=apply_view,
                                         diag_scale=diag_scale)
    vis = torch.as_tensor(disc['radii'] > 0)
    # culled gaussians keep zero centre/colour (forward_pass.py:122-127); they own no tiles
    centers = torch.where(vis[:, None], centers, torch.zeros_like(centers))
    colors = torch.where(vis[:, None], colors, torch.zeros_like(colors))
    conics = torch.where(vis[:, None], conics, torch.zeros_like(conics))
    img = blend(centers, colors, opacities, conics, vals, ranges, result_size, tile_size,
                background, pixel_offset)
    return img, dict(disc=disc, vals=vals, ranges=ranges, centers=centers, conics=conics,
                     colors=colors)
