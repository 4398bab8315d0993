"""ORACLE (test infrastructure, never the product path).

NumPy fp64 restatement of the reference's CPU hot path, vectorised over gaussians:

  preprocess()        <- forward_pass.py:25-136 + py_compute_color (forward_pass.py:168-215)
                          + helpers functions.py:4-45
  tile_match_dense()  <- setup_gpu.py:23-44 (kernel tile_match), rasterization.py:6-25
  per_tile_lists_dense() <- rasterization.py:66-70 (filter low32 != 0, stable sort on low 32 bits)
  bin_duplicates()    <- the O(D) equivalent of the two above: (tile<<32 | depth) keys, stable sort
  blend_forward_py()  <- rasterization.py:155-215 / rasterize.c:82-131 (pure-Python, tiny cases)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against golden vectors captured from the reference itself (oracle/make_golden.py ->
tests/golden/*.npz); see tests/test_oracle_golden.py.

Third-party semantics restated here because the packages are absent from the image
(requirements.txt:2,6 are unpinned):
  numpy-quaternion  as_quat_array/as_rotation_matrix: [w,x,y,z], matrix divided by the squared
                    norm (call sites forward_pass.py:74, backward_pass.py:52)
  pyopencl RadixSort(key_bits=32): stable LSD sort on the low 32 bits (rasterization.py:57,70)
"""
import math
import numpy as np

# forward_pass.py:8-21
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = np.array([1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                  -1.0925484305920792, 0.5462742152960396])
SH_C3 = np.array([-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                  0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                  -0.5900435899266435])


# ---------------------------------------------------------------- functions.py helpers
def focal_to_fov(focal_len, pixels):          # functions.py:36-37
    return 2.0 * math.atan(pixels / (2.0 * focal_len))


def get_view_matrix(r, t):                     # functions.py:4-9
    v = np.zeros([4, 4])
    v[:3, :3] = np.transpose(r)
    v[:3, 3] = t
    v[3, 3] = 1.0
    return v


def get_projection_matrix(fov_x, fov_y, z_near=.01, z_far=100):   # functions.py:11-22
    top = math.tan(fov_y / 2) * z_near
    right = math.tan(fov_x / 2) * z_near
    p = np.zeros([4, 4])
    p[0, 0] = z_near / right
    p[1, 1] = z_near / top
    p[2, 2] = z_far / (z_far - z_near)
    p[2, 3] = -(z_far * z_near) / (z_far - z_near)
    p[3, 2] = 1.0
    return p


def get_pixel(num, scale):                     # functions.py:38-39
    return ((num + 1.0) * scale - 1.0) * .5


def sigmoid(x):                                # functions.py:42-43
    return 1 / (1 + np.exp(-x))


def quat_to_rotmat(q):
    """numpy-quaternion as_rotation_matrix on [w,x,y,z] rows (divides by the squared norm)."""
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s = 2.0 / (w * w + x * x + y * y + z * z)
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - s * (y * y + z * z)
    R[..., 0, 1] = s * (x * y - z * w)
    R[..., 0, 2] = s * (x * z + y * w)
    R[..., 1, 0] = s * (x * y + z * w)
    R[..., 1, 1] = 1 - s * (x * x + z * z)
    R[..., 1, 2] = s * (y * z - x * w)
    R[..., 2, 0] = s * (x * z - y * w)
    R[..., 2, 1] = s * (y * z + x * w)
    R[..., 2, 2] = 1 - s * (x * x + y * y)
    return R


def sh_basis(dirs, degree):
    """Basis values b[N,16] so that colour[ch] = sum_k b[k]*sh[ch][k] (forward_pass.py:173-204)."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    b = np.zeros((len(dirs), 16))
    b[:, 0] = SH_C0
    if degree > 0:
        b[:, 1] = -y * SH_C1
        b[:, 2] = z * SH_C1
        b[:, 3] = -x * SH_C1
        if degree > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            b[:, 4] = SH_C2[0] * xy
            b[:, 5] = SH_C2[1] * yz
            b[:, 6] = SH_C2[2] * (2.0 * zz - xx - yy)
            b[:, 7] = SH_C2[3] * xz
            b[:, 8] = SH_C2[4] * (xx - yy)
            if degree > 2:
                b[:, 9] = SH_C3[0] * y * (3.0 * xx - yy)
                b[:, 10] = SH_C3[1] * xy * z
                b[:, 11] = SH_C3[2] * y * (4.0 * zz - xx - yy)
                b[:, 12] = SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy)
                b[:, 13] = SH_C3[4] * x * (4.0 * zz - xx - yy)
                b[:, 14] = SH_C3[5] * z * (xx - yy)
                b[:, 15] = SH_C3[6] * x * (xx - 3.0 * yy)
    return b


def _py_clamp_to_tiles(v, size, tile):
    """int(min(size, max(0, v)) / tile) with Python's min/max NaN behaviour
    (forward_pass.py:116-119): max(0, nan) -> 0."""
    m = np.where(v > 0, v, 0.0)          # Python max(0, v): v only if v > 0
    m = np.where(m < size, m, float(size))   # Python min(size, m): m only if m < size
    return np.trunc(m / tile).astype(np.int32)


def preprocess(means, log_scales, quats, sh, degree, fx, fy, cam_width, cam_height,
               camera_r, camera_t, result_size=(979, 546), tile_size=16, apply_view=False, diag_scale=False):
    """forward_pass.py:25-136 (training=True outputs), vectorised.

    apply_view / diag_scale (default False = the reference): the library's GS_CAM_APPLY_VIEW /
    GS_CAM_DIAG_SCALE variants -- project the VIEW-space position (the line the reference has commented
    out, forward_pass.py:57) / zero the off-diagonals of S.  These two have no reference output to pin
    them: "parity unpinned", checked only oracle-vs-library.

    means[N,3], log_scales[N,3], quats[N,4] (w,x,y,z), sh[N,3,16]; camera_r[3,3], camera_t[3].
    Returns a dict with the 13 reference outputs (same names as forward_pass.py:134 locals,
    minus the 'mod_' prefix).
    """
    means = np.asarray(means, dtype=np.float64)
    log_scales = np.asarray(log_scales, dtype=np.float64)
    quats = np.asarray(quats, dtype=np.float64)
    sh = np.asarray(sh, dtype=np.float64)
    camera_r = np.asarray(camera_r, dtype=np.float64)
    camera_t = np.asarray(camera_t, dtype=np.float64)
    N = len(means)
    W, H = int(result_size[0]), int(result_size[1])

    fov_x = focal_to_fov(fx, cam_width)          # forward_pass.py:43
    fov_y = focal_to_fov(fx, cam_height)         # forward_pass.py:44 (fx for both: quirk Q2)
    proj = get_projection_matrix(fov_x, fov_y)
    view = get_view_matrix(camera_r, camera_t)
    cam_pos = np.transpose(camera_r) @ camera_t  # forward_pass.py:47 (quirk Q3)
    scales = np.exp(log_scales)                  # :49
    scale_mod = (1.0 / H) * 3                    # :50

    out = dict(
        centers=np.zeros((N, 2)), depths=np.zeros(N, dtype=np.int32), colors=np.zeros((N, 3)),
        conics=np.zeros((N, 3)), clamped=np.zeros((N, 3)),
        tiles_touched=np.zeros((N, 2, 2), dtype=np.int32), radii=np.zeros(N),
        Ws=np.zeros((N, 6)), Ms=np.zeros((N, 3, 3)), Ts=np.zeros((N, 3)),
        cov2d=np.zeros((N, 3)), cov3d=np.zeros((N, 6)), dirs=np.zeros((N, 3)))
    if N == 0:
        return out

    orig = np.concatenate([means, np.ones((N, 1))], axis=1)
    with np.errstate(all='ignore'):
        vc = orig @ view.T                                       # :64
        if apply_view:
            vh = np.concatenate([vc[:, :3], np.ones((N, 1))], axis=1)
            pc = vh @ proj.T                                     # :57 (commented out in the reference)
        else:
            pc = orig @ proj.T                                   # :59
        mod_w = 1.0 / (pc[:, 3] + .00000001)                     # :60
        center_ndc = pc * mod_w[:, None]                         # :61
        zv = vc[:, 2]
        alive = ~((zv <= 2) | np.isnan(zv))                      # :65-67

        # depth key (:70-71): trunc toward zero, wrapped to int32
        d = np.trunc(np.where(alive, zv, 0.0) * 1000.0)
        d = np.where(np.isfinite(d), d, 0.0)
        d64 = d.astype(np.float64)
        wrapped = np.mod(d64 + 2147483648.0, 4294967296.0) - 2147483648.0
        out['depths'][:] = np.where(alive, wrapped, 0).astype(np.int64).astype(np.int32)

        qn = quats / np.linalg.norm(quats, axis=1, keepdims=True)   # :74
        R = quat_to_rotmat(qn)
        S = np.zeros((N, 3, 3)) if diag_scale else np.ones((N, 3, 3))
        S[:, 0, 0], S[:, 1, 1], S[:, 2, 2] = scales[:, 0], scales[:, 1], scales[:, 2]
        S = S * scale_mod                                        # :75 (off-diagonals are 1: quirk Q4)
        M = R @ S
        cov3 = M @ np.transpose(M, (0, 2, 1))                    # :76

        # Jacobian (functions.py:24-34, returned transposed) and W = J^T @ camera_r^T (:78-80)
        t = vc
        lim_x = 1.3 * math.tan(fov_x * .5)
        lim_y = 1.3 * math.tan(fov_y * .5)
        mx = np.minimum(lim_x, np.maximum(-lim_x, t[:, 0] / t[:, 2])) * t[:, 2]
        my = np.minimum(lim_y, np.maximum(-lim_y, t[:, 1] / t[:, 2])) * t[:, 2]
        mz = t[:, 2]
        Jt = np.zeros((N, 3, 3))
        Jt[:, 0, 0] = fx / mz
        Jt[:, 1, 1] = fy / mz
        Jt[:, 2, 0] = -(fx * mx) / (mz * mz)
        Jt[:, 2, 1] = -(fy * my) / (mz * mz)
        Wm = Jt @ np.transpose(camera_r)                          # :80
        c2 = Wm @ cov3 @ np.transpose(Wm, (0, 2, 1))              # :81
        a = c2[:, 0, 0] + .3                                      # :85
        b = c2[:, 0, 1]
        c = c2[:, 1, 1] + .3                                      # :86

        al = alive
        out['Ms'][al] = M[al]
        out['Ws'][al] = np.stack([Wm[:, 0, 0], Wm[:, 0, 1], Wm[:, 0, 2],
                                  Wm[:, 1, 0], Wm[:, 1, 1], Wm[:, 1, 2]], axis=1)[al]   # :91
        out['Ts'][al] = t[al, :3]
        out['cov2d'][al] = np.stack([a, b, c], axis=1)[al]
        out['cov3d'][al] = np.stack([cov3[:, 0, 0], cov3[:, 0, 1], cov3[:, 0, 2],
                                     cov3[:, 1, 1], cov3[:, 1, 2], cov3[:, 2, 2]], axis=1)[al]

        det = a * c - b * b                                       # :97
        al = al & ~(det == 0)                                     # :98-99
        det_inv = 1.0 / det
        conic = np.stack([c * det_inv, -b * det_inv, a * det_inv], axis=1)   # :101
        out['conics'][al] = conic[al]

        mid = .5 * (a + c)                                        # :106
        disc = np.sqrt(np.maximum(.1, mid * mid - det))
        l1, l2 = mid + disc, mid - disc
        # Python max(l1, l2, 0): NaN-propagation differs from np.maximum only for NaN inputs,
        # which are culled on the next line anyway
        val = np.sqrt(np.maximum(np.maximum(l1, l2), 0.0))        # :109
        al = al & ~np.isnan(val)                                  # :110-111
        radius = np.ceil(3 * np.where(np.isnan(val), 0.0, val))   # :112

        cx = get_pixel(center_ndc[:, 0], W)                       # :114
        cy = get_pixel(center_ndc[:, 1], H)
        tt = np.zeros((N, 2, 2), dtype=np.int32)
        tt[:, 0, 0] = _py_clamp_to_tiles(cx - radius, W, tile_size)
        tt[:, 0, 1] = _py_clamp_to_tiles(cy - radius, H, tile_size)
        tt[:, 1, 0] = _py_clamp_to_tiles(cx + radius + tile_size - 1, W, tile_size)
        tt[:, 1, 1] = _py_clamp_to_tiles(cy + radius + tile_size - 1, H, tile_size)
        out['tiles_touched'][al] = tt[al]                         # written before the cull (Q10)

        off = (cx + radius < 0) | (cx - radius > W) | (cy + radius < 0) | (cy - radius > H)   # :122
        al = al & ~off
        out['radii'][al] = radius[al]
        out['centers'][al] = np.stack([cx, cy], axis=1)[al]

        # py_compute_color (:168-215)
        dvec = means - cam_pos
        dirs = dvec / np.linalg.norm(dvec, axis=1, keepdims=True)
        basis = sh_basis(dirs, degree)
        col = np.einsum('nk,nck->nc', basis, sh) + .5
        cl = col < 0
        col = np.where(cl, 0.0, col)
        out['colors'][al] = col[al]
        out['clamped'][al] = cl[al].astype(np.float64)
        out['dirs'][al] = dirs[al]
    return out


# ---------------------------------------------------------------- binning
def tile_grid(result_size, tile_size):
    return (math.ceil(result_size[0] * 1.0 / tile_size), math.ceil(result_size[1] * 1.0 / tile_size))


def tile_match_dense(tiles_touched, depths, result_size, tile_size=16):
    """Literal restatement of kernel tile_match (setup_gpu.py:23-44): dense u64
    key_mapper[tiles_x, tiles_y, N] = (g<<32) + (depth if tile in rect else 0).
    O(tiles*N) memory -- small cases only."""
    tx_n, ty_n = tile_grid(result_size, tile_size)
    tt = np.asarray(tiles_touched, dtype=np.int64).reshape(-1, 4)
    N = len(tt)
    gx = np.arange(tx_n)[:, None, None]
    gy = np.arange(ty_n)[None, :, None]
    inside = ((gx >= tt[None, None, :, 0]) & (gx < tt[None, None, :, 2]) &
              (gy >= tt[None, None, :, 1]) & (gy < tt[None, None, :, 3]))
    g = (np.arange(N, dtype=np.uint64) << np.uint64(32))[None, None, :]
    # (ulong)key + int depth: the int is converted to ulong (sign-extended) before the add
    dep = np.asarray(depths, dtype=np.int64).astype(np.uint64)[None, None, :]
    return np.where(inside, g + dep, g + np.uint64(0)).astype(np.uint64)


def per_tile_lists_dense(key_mapper):
    """rasterization.py:66-70: keep keys whose low 32 bits are non-zero, stable sort on the low
    32 bits (unsigned). Returns dict[(tx,ty)] -> u64 array (the reference's gaus_stack)."""
    out = {}
    tx_n, ty_n, _ = key_mapper.shape
    for i in range(tx_n):
        for j in range(ty_n):
            k = key_mapper[i, j]
            k = k[(k << np.uint64(32)) != 0]
            order = np.argsort(k & np.uint64(0xFFFFFFFF), kind='stable')
            out[(i, j)] = k[order]
    return out


def bin_duplicates(tiles_touched, depths, result_size, tile_size=16):
    """O(D) equivalent of tile_match + per-tile filter + stable sort.

    Emits one (key, val) per (gaussian, tile in rect) with key = (tile_id<<32) | u32(depth),
    tile_id = tx*tiles_y + ty (the key_mapper[grid_x][grid_y] order, setup_gpu.py:28-30), val = g;
    entries with u32(depth) == 0 are dropped (rasterization.py:67); stable sort on the key with
    the input in gaussian-ascending order => per tile (depth asc, gaussian asc) (quirk Q12).
    Returns keys u64[D], vals i32[D], ranges i32[tiles,2] ([start,end) per tile_id).
    """
    tx_n, ty_n = tile_grid(result_size, tile_size)
    tt = np.asarray(tiles_touched, dtype=np.int64).reshape(-1, 4)
    N = len(tt)
    x0 = np.clip(tt[:, 0], 0, tx_n); x1 = np.clip(tt[:, 2], 0, tx_n)
    y0 = np.clip(tt[:, 1], 0, ty_n); y1 = np.clip(tt[:, 3], 0, ty_n)
    d32 = np.asarray(depths, dtype=np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    w = np.maximum(x1 - x0, 0); h = np.maximum(y1 - y0, 0)
    cnt = np.where(d32 != 0, w * h, 0)
    D = int(cnt.sum())
    g = np.repeat(np.arange(N, dtype=np.int64), cnt)
    first = np.cumsum(cnt) - cnt
    local = np.arange(D, dtype=np.int64) - np.repeat(first, cnt)
    hh = np.repeat(h, cnt)
    hh = np.where(hh == 0, 1, hh)
    tx = np.repeat(x0, cnt) + local // hh
    ty = np.repeat(y0, cnt) + local % hh
    tile = (tx * ty_n + ty).astype(np.uint64)
    keys = (tile << np.uint64(32)) | np.repeat(d32, cnt)
    order = np.argsort(keys, kind='stable')
    keys = keys[order]
    vals = g[order].astype(np.int32)
    tiles = tx_n * ty_n
    ranges = np.zeros((tiles, 2), dtype=np.int32)
    if D:
        t_sorted = (keys >> np.uint64(32)).astype(np.int64)
        starts = np.searchsorted(t_sorted, np.arange(tiles), side='left')
        ends = np.searchsorted(t_sorted, np.arange(tiles), side='right')
        nz = ends > starts
        ranges[nz, 0] = starts[nz]
        ranges[nz, 1] = ends[nz]
    return keys, vals, ranges


def lists_from_sorted(keys, vals, ranges, result_size, tile_size=16):
    """Materialise the reference's per-tile gaus_stack arrays ((g<<32)+depth, u64) from the
    (tile<<32|depth, g) representation, for bit-exact comparison with per_tile_lists_dense()."""
    tx_n, ty_n = tile_grid(result_size, tile_size)
    out = {}
    for i in range(tx_n):
        for j in range(ty_n):
            s, e = ranges[i * ty_n + j]
            k = keys[s:e] & np.uint64(0xFFFFFFFF)
            # undo the u32 truncation the same way the reference's add would have produced it
            out[(i, j)] = (vals[s:e].astype(np.uint64) << np.uint64(32)) + k
    return out


# ---------------------------------------------------------------- blend (pure Python, tiny)
def blend_forward_py(centers, colors, opacities, conics, vals, ranges, result_size,
                     tile_size=16, background=(0, 0, 0), pixel_offset=0.5):
    """rasterize.c:82-131 loop (pixel_offset=+0.5: dist = c - x + .5) /
    rasterization.py:171-212 (pixel_offset=-0.5: dist = c - (x + .5)) /
    setup_gpu.py:70-100 (pixel_offset=0). Returns image[W,H,3], last_contributors[W,H], T[W,H]."""
    W, H = result_size
    tx_n, ty_n = tile_grid(result_size, tile_size)
    bg = np.asarray(background, dtype=np.float64)
    image = np.zeros((W, H, 3)); contrib = np.zeros((W, H), dtype=np.int32); Tf = np.ones((W, H))
    for i in range(tx_n):
        for j in range(ty_n):
            s, e = ranges[i * ty_n + j]
            lst = vals[s:e]
            for x in range(i * tile_size, min((i + 1) * tile_size, W)):
                for y in range(j * tile_size, min((j + 1) * tile_size, H)):
                    T = 1.0; col = np.zeros(3); n = 0
                    for g in lst:
                        n += 1
                        dx = centers[g, 0] - x + pixel_offset
                        dy = centers[g, 1] - y + pixel_offset
                        cn = conics[g]
                        power = -.5 * (cn[0] * dx * dx + cn[2] * dy * dy) - cn[1] * dx * dy
                        if power > 0:
                            continue
                        alpha = min(.99, opacities[g] * math.exp(power))
                        if alpha < (1.0 / 255):
                            continue
                        test_T = T * (1 - alpha)
                        if test_T < .0001:
                            break
                        col += colors[g] * alpha * T
                        T = test_T
                    image[x, y] = col + bg * T
                    contrib[x, y] = n
                    Tf[x, y] = T
    return image, contrib, Tf


# ---------------------------------------------------------------- the device's own branch decisions, for gradient tests
def _fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64, one rounding at the end."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def device_decisions(centers, conics, opacities, vals, ranges, result_size, tile_size, pixel_offset, n_contrib):
    """The branches K6 / K7 took, for ref_torch.blend(forced=...): a gradient oracle that follows them can be
    compared on EVERY pixel, the ones included where fp32 lands on the other side of one of the reference's
    discontinuities (rasterize.c:109-119: power > 0, alpha < 1/255, T < 1e-4).

    From the device's output n_contrib[W,H] (bits 0..30 = list entries visited, the breaking one included =
    rasterize.c's last_contributors; bit 31 = left through the transmittance break) and from the validity of
    every (pixel, list entry) pair recomputed here in float32 with the arithmetic of the kernels (csrc/blend.hip,
    blend_fwd_quad_kernel: log2-scaled conic, the far pixels of a 16x16 block 8 px on by one subtraction,
    p = fma(dx, fma(b', dy, a' dx), (c' dy) dy), valid = p <= 0 and not(min(.99, o * 2^p) < 1/255)).
    Returns dict(visited[tiles,P], broke[tiles,P], valid[tiles,L,P]) in ref_torch.blend's layout (P = tile_size^2,
    pixel lx * tile_size + ly)."""
    W, H = int(result_size[0]), int(result_size[1])
    ts = int(tile_size)
    tx_n, ty_n = tile_grid(result_size, ts)
    tiles, P = tx_n * ty_n, ts * ts
    f32 = np.float32
    cen, con, opa = np.asarray(centers, f32), np.asarray(conics, f32), np.asarray(opacities, f32).reshape(-1)
    LOG2E = f32(1.4426950408889634)
    a2 = (f32(-0.5) * LOG2E) * con[:, 0]
    b2 = (-LOG2E) * con[:, 1]
    c2 = (f32(-0.5) * LOG2E) * con[:, 2]
    off = f32(pixel_offset)
    ranges = np.asarray(ranges).reshape(tiles, 2)
    lens = ranges[:, 1] - ranges[:, 0]
    L = int(lens.max()) if tiles else 0
    lx = np.repeat(np.arange(ts), ts)
    ly = np.tile(np.arange(ts), ts)
    nc = np.asarray(n_contrib).astype(np.int64)
    visited = np.zeros((tiles, P), dtype=np.int64)
    broke = np.zeros((tiles, P), dtype=bool)
    valid = np.zeros((tiles, L, P), dtype=bool)
    for t in range(tiles):
        tx, ty = t // ty_n, t % ty_n
        px, py = tx * ts + lx, ty * ts + ly
        inside = (px < W) & (py < H)
        v = nc[np.minimum(px, W - 1), np.minimum(py, H - 1)]
        visited[t] = np.where(inside, v & 0x7FFFFFFF, 0)
        broke[t] = inside & ((v >> 31) & 1).astype(bool)
        n = int(lens[t])
        if n == 0:
            continue
        g = np.asarray(vals[ranges[t, 0]:ranges[t, 1]]).astype(np.int64)
        # near pixel of the lane's quadrant, then one subtraction of 8 for the far ones
        bx, by = (px // 16) * 16, (py // 16) * 16
        fx0 = (bx + px % 8).astype(f32) - off
        fy0 = (by + py % 8).astype(f32) - off
        dx = cen[g, 0][:, None] - fx0[None, :]
        dy = cen[g, 1][:, None] - fy0[None, :]
        dx = np.where((px % 16 >= 8)[None, :], dx - f32(8.0), dx).astype(f32)
        dy = np.where((py % 16 >= 8)[None, :], dy - f32(8.0), dy).astype(f32)
        adx = (a2[g][:, None] * dx).astype(f32)
        cdy = (c2[g][:, None] * dy).astype(f32)
        p = _fma32(dx, _fma32(np.broadcast_to(b2[g][:, None], dx.shape), dy, adx), (cdy * dy).astype(f32))
        with np.errstate(over="ignore", invalid="ignore"):
            G = np.exp2(p.astype(np.float64)).astype(f32)
            alpha = np.minimum(f32(0.99), (opa[g][:, None] * G).astype(f32))
            valid[t, :n] = (p <= 0) & ~(alpha < f32(1.0 / 255.0))
    return dict(visited=visited, broke=broke, valid=valid)
