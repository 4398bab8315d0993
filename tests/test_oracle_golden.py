"""CPU: the oracle restatements against the golden vectors captured from the reference itself
(oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import load_scene
from oracle import cref, ref_np, ref_torch

SCENES = ["s0", "s1", "s2", "s3", "s4"]
IMG_SCENES = ["s0", "s1", "s2", "s3"]


def _pre(d):
    return ref_np.preprocess(d["in_center"], d["in_scaling"], d["in_rotation"], d["in_sh"], int(d["in_degrees"]),
                             float(d["in_fx"]), float(d["in_fy"]), int(d["in_cam_w"]), int(d["in_cam_h"]),
                             d["in_R"], d["in_t"], d["in_result_size"].tolist(), int(d["in_tile_size"]))


@pytest.mark.parametrize("name", SCENES)
def test_preprocess_matches_reference(name):
    d = load_scene(name)
    o = _pre(d)
    for k, v in o.items():
        g = d["fp_" + k]
        assert g.shape == v.shape, k
        if g.dtype.kind == "i":
            assert np.array_equal(g, v), k           # depth keys, tile rects: bit-exact
        else:
            np.testing.assert_allclose(v, g, rtol=1e-12, atol=1e-12, err_msg=k)
    assert np.array_equal(o["radii"], d["fp_radii"])  # integer-valued floats


def test_smoke_values_of_the_survey():
    d = load_scene("s0")
    assert d["fp_radii"][:8].tolist() == [5, 8, 7, 6, 6, 6, 7, 6]
    assert d["fp_depths"][:4].tolist() == [3622, 4441, 5930, 5770]
    assert abs(d["fp_centers"][0, 0] - 172.8669055) < 1e-6


def test_edge_gaussians_are_culled_like_the_reference():
    d = load_scene("s2")
    o = _pre(d)
    assert o["radii"][0] == 0 and o["radii"][2] == 0     # off-screen (Q10)
    assert o["depths"][1] == 0 and o["depths"][3] == 0   # z <= 2 (Q6), incl. z == 2 exactly
    tt = o["tiles_touched"]
    for g in (0, 2):   # off-screen rect is empty on the culled axis
        assert tt[g, 0, 0] == tt[g, 1, 0] or tt[g, 0, 1] == tt[g, 1, 1]


@pytest.mark.parametrize("name", SCENES)
def test_binning_equals_literal_tile_match(name):
    d = load_scene(name)
    rs = d["in_result_size"].tolist()
    keys, vals, ranges = ref_np.bin_duplicates(d["fp_tiles_touched"], d["fp_depths"], rs, 16)
    dense = ref_np.per_tile_lists_dense(ref_np.tile_match_dense(d["fp_tiles_touched"], d["fp_depths"], rs, 16))
    fast = ref_np.lists_from_sorted(keys, vals, ranges, rs, 16)
    assert dense.keys() == fast.keys()
    for k in dense:
        assert np.array_equal(dense[k], fast[k]), k
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)
    assert int((ranges[:, 1] - ranges[:, 0]).sum()) == len(keys)


def test_ties_resolve_by_gaussian_id():
    d = load_scene("s4")   # four depth planes -> many equal keys per tile (Q12)
    rs = d["in_result_size"].tolist()
    keys, vals, ranges = ref_np.bin_duplicates(d["fp_tiles_touched"], d["fp_depths"], rs, 16)
    ties = 0
    for s, e in ranges:
        k, v = keys[s:e], vals[s:e]
        same = k[1:] == k[:-1]
        ties += int(same.sum())
        assert np.all(v[1:][same] > v[:-1][same])
    assert ties > 100


def test_q9_last_partial_tile_row_is_empty():
    d = load_scene("s3")   # 203x117: x-tile 12 and y-tile 7 are partial
    rs = d["in_result_size"].tolist()
    _, _, ranges = ref_np.bin_duplicates(d["fp_tiles_touched"], d["fp_depths"], rs, 16)
    tx_n, ty_n = ref_np.tile_grid(rs, 16)
    r = ranges.reshape(tx_n, ty_n, 2)
    assert np.all(r[tx_n - 1, :, 1] - r[tx_n - 1, :, 0] == 0)
    assert np.all(r[:, ty_n - 1, 1] - r[:, ty_n - 1, 0] == 0)


@pytest.mark.parametrize("name", IMG_SCENES)
def test_c_blend_matches_reference_images(name):
    d = load_scene(name)
    rs = d["in_result_size"].tolist()
    _, vals, ranges = ref_np.bin_duplicates(d["fp_tiles_touched"], d["fp_depths"], rs, 16)
    img, lc, tf, fr, _ = cref.rasterize_frame(d["fp_centers"], d["fp_colors"], d["in_opacity"], d["fp_conics"],
                                              vals, ranges, rs, d["in_bg"], pixel_offset=-0.5)
    np.testing.assert_allclose(img, d["img"], rtol=0, atol=1e-15)


def test_python_and_c_blend_agree_on_contributors():
    d = load_scene("s0")
    rs = d["in_result_size"].tolist()
    _, vals, ranges = ref_np.bin_duplicates(d["fp_tiles_touched"], d["fp_depths"], rs, 16)
    a = ref_np.blend_forward_py(d["fp_centers"], d["fp_colors"], d["in_opacity"], d["fp_conics"], vals, ranges,
                                rs, 16, d["in_bg"], 0.5)
    b = cref.rasterize_frame(d["fp_centers"], d["fp_colors"], d["in_opacity"], d["fp_conics"], vals, ranges, rs,
                             d["in_bg"], pixel_offset=0.5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("name", ["s1", "s3"])
def test_torch_oracle_forward_matches_reference_image(name):
    import torch
    d = load_scene(name)
    T = lambda k: torch.tensor(d[k], dtype=torch.float64)
    img, _ = ref_torch.render(T("in_center"), T("in_scaling"), T("in_rotation"), T("in_sh"), T("in_opacity"),
                              int(d["in_degrees"]), float(d["in_fx"]), float(d["in_fy"]), int(d["in_cam_w"]),
                              int(d["in_cam_h"]), d["in_R"], d["in_t"], d["in_result_size"].tolist(), 16,
                              d["in_bg"], -0.5)
    np.testing.assert_allclose(img.numpy(), d["img"], rtol=0, atol=1e-13)


def test_torch_oracle_gradient_against_finite_differences():
    """The gradient oracle itself: autograd vs central differences on a few parameters."""
    import torch
    d = load_scene("s0")
    rs = d["in_result_size"].tolist()
    rng = np.random.default_rng(5)
    wimg = torch.tensor(rng.normal(size=(rs[0], rs[1], 3)))
    base = {k: torch.tensor(d[k], dtype=torch.float64) for k in
            ("in_center", "in_scaling", "in_rotation", "in_sh", "in_opacity")}

    def loss(p):
        img, _ = ref_torch.render(p["in_center"], p["in_scaling"], p["in_rotation"], p["in_sh"], p["in_opacity"],
                                  3, 300., 300., 256, 256, d["in_R"], d["in_t"], rs, 16, d["in_bg"], 0.5)
        return (img * wimg).sum()

    p = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    loss(p).backward()
    vis = np.nonzero(d["fp_radii"] > 0)[0]
    for key, idx in (("in_center", (vis[3], 0)), ("in_center", (vis[5], 2)), ("in_scaling", (vis[2], 1)),
                     ("in_rotation", (vis[4], 2)), ("in_sh", (vis[1], 1, 5)), ("in_opacity", (vis[6],))):
        h = 1e-6
        q = {k: v.clone() for k, v in base.items()}
        q[key][idx] += h
        up = loss(q).item()
        q[key][idx] -= 2 * h
        dn = loss(q).item()
        fd = (up - dn) / (2 * h)
        an = p[key].grad[idx].item()
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (key, idx, fd, an)


@pytest.mark.parametrize("apply_view,diag_scale", [(True, False), (False, True), (True, True)])
def test_sane_mode_oracles_agree_with_each_other(apply_view, diag_scale):
    """The apply_view / diag_scale variants (no reference counterpart: "parity unpinned") -- the numpy and
    the torch restatement agree on the continuous outputs, the defaults still give the golden values, and
    autograd through the variant matches central differences."""
    import torch
    d = load_scene("s1")
    rs = d["in_result_size"].tolist()
    args = (int(d["in_degrees"]), float(d["in_fx"]), float(d["in_fy"]), int(d["in_cam_w"]), int(d["in_cam_h"]),
            d["in_R"], d["in_t"], rs)
    modes = dict(apply_view=apply_view, diag_scale=diag_scale)
    o = ref_np.preprocess(d["in_center"], d["in_scaling"], d["in_rotation"], d["in_sh"], *args, 16, **modes)
    T = lambda k: torch.tensor(d[k], dtype=torch.float64)
    c, con, col = ref_torch.preprocess(T("in_center"), T("in_scaling"), T("in_rotation"), T("in_sh"), *args, **modes)
    vis = o["radii"] > 0
    assert vis.sum() >= 10
    np.testing.assert_allclose(c.numpy()[vis], o["centers"][vis], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(con.numpy()[vis], o["conics"][vis], rtol=1e-9, atol=1e-12)
    changed = (not np.allclose(o["centers"], d["fp_centers"])) or (not np.allclose(o["conics"], d["fp_conics"]))
    assert changed
    if diag_scale:
        M = o["Ms"][vis]
        S2 = np.einsum("nij,nik->njk", M, M)              # S^T R^T R S = S^2: diagonal
        off = S2 - np.einsum("nii->ni", S2)[:, :, None] * np.eye(3)
        assert np.abs(off).max() < 1e-12

    p = T("in_center").clone().requires_grad_(True)
    w = torch.tensor(np.random.default_rng(2).normal(size=(int(vis.sum()), 2)))
    f = lambda x: (ref_torch.preprocess(x, T("in_scaling"), T("in_rotation"), T("in_sh"), *args, **modes)[0][vis] * w).sum()
    f(p).backward()
    i = int(np.nonzero(vis)[0][3])
    for k in range(3):
        q = T("in_center").clone()
        q[i, k] += 1e-6
        up = f(q).item()
        q[i, k] -= 2e-6
        fd = (up - f(q).item()) / 2e-6
        assert abs(fd - p.grad[i, k].item()) <= 1e-5 * max(1.0, abs(fd))
