"""CPU: data formats either side of the hot path -- the COLMAP
reader against fixtures written by the reference's own writer, the 3DGS PLY reader against a
byte-level fixture built from the format description, the npz checkpoint layout of
gaussian.py:54-56, and the kNN scale initialisation against values computed by the reference."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

COLMAP = os.path.join(GOLDEN, "colmap")


@pytest.mark.parametrize("ext", [".bin", ".txt", ""])
def test_colmap_reader_matches_reference_written_model(ext):
    import gsplat_amd as gs
    exp = np.load(os.path.join(COLMAP, "expected.npz"), allow_pickle=False)
    cameras, images, points = gs.read_model(COLMAP, ext)
    assert sorted(cameras) == exp["camera_ids"].tolist()
    assert sorted(images) == exp["image_ids"].tolist()
    assert sorted(points) == exp["point_ids"].tolist()
    tol = dict(rtol=0, atol=0) if ext != ".txt" else dict(rtol=1e-12, atol=1e-12)
    for cid, c in cameras.items():
        np.testing.assert_allclose(c.params, exp[f"cam{cid}_params"], **tol)
        assert [c.width, c.height] == exp[f"cam{cid}_wh"].tolist() and c.model == str(exp[f"cam{cid}_model"])
    for iid, im in images.items():
        np.testing.assert_allclose(im.qvec, exp[f"img{iid}_qvec"], **tol)
        np.testing.assert_allclose(im.tvec, exp[f"img{iid}_tvec"], **tol)
        assert im.camera_id == int(exp[f"img{iid}_cam"]) and im.name == str(exp[f"img{iid}_name"])
        np.testing.assert_allclose(im.xys.reshape(-1, 2), exp[f"img{iid}_xys"], **tol)
        assert np.array_equal(im.point3D_ids, exp[f"img{iid}_p3d"])
        np.testing.assert_allclose(gs.qvec2rotmat(im.qvec), exp[f"img{iid}_R"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(gs.quat_to_rotmat(im.qvec), exp[f"img{iid}_R"], rtol=1e-12, atol=1e-14)
    for pid, p in points.items():
        np.testing.assert_allclose(p.xyz, exp[f"pt{pid}_xyz"], **tol)
        assert np.array_equal(p.rgb, exp[f"pt{pid}_rgb"])
        np.testing.assert_allclose(p.error, exp[f"pt{pid}_err"], **tol)
        assert np.array_equal(p.image_ids, exp[f"pt{pid}_img"]) and np.array_equal(p.point2D_idxs, exp[f"pt{pid}_idx"])


def test_colmap_reader_rejects_truncated_files(tmp_path):
    import gsplat_amd as gs
    raw = open(os.path.join(COLMAP, "images.bin"), "rb").read()
    for name in ("cameras.bin", "points3D.bin"):
        (tmp_path / name).write_bytes(open(os.path.join(COLMAP, name), "rb").read())
    (tmp_path / "images.bin").write_bytes(raw[:len(raw) - 9])
    with pytest.raises(ValueError):
        gs.read_model(str(tmp_path), ".bin")
    with pytest.raises(FileNotFoundError):
        gs.read_model(str(tmp_path / "missing"), "")


def test_knn_scale_init_matches_reference_values():
    import gsplat_amd as gs
    exp = np.load(os.path.join(COLMAP, "expected.npz"), allow_pickle=False)
    np.testing.assert_allclose(gs.knn_distances(exp["knn_centers"]), exp["knn_distances"], rtol=1e-13)
    _, _, points = gs.read_model(COLMAP, ".bin")
    g = gs.gaussians_from_point_cloud(points)
    n = len(points)
    assert g.center.shape == (n, 3) and g.spherical_harmonics.shape == (n, 3, 16) and g.degrees == 0
    order = sorted(points)
    assert list(points) == order            # the reader keeps file order = id order of the fixture
    np.testing.assert_allclose(g.scaling[:, 0], np.log(np.clip(exp["knn_distances"], 1e-7, None)), rtol=1e-13)
    assert np.all(g.scaling[:, 0:1] == g.scaling) and np.all(g.opacity == .1)
    assert np.all(g.rotation == np.array([0., 0., 0., 1.]))
    rgb = np.array([points[k].rgb for k in order], dtype=np.float64)
    np.testing.assert_allclose(g.spherical_harmonics[:, :, 0], (rgb / 255 - .5) / 0.28209479177387814)
    assert np.all(g.spherical_harmonics[:, :, 1:] == 0)
    g1 = gs.gaussians_from_point_cloud(points, rgb_scale=1.0)     # gaussian.py:19 verbatim
    np.testing.assert_allclose(g1.spherical_harmonics[:, :, 0], (rgb - .5) / 0.28209479177387814)


def _handmade_ply(path, n, rng, fmt="binary_little_endian"):
    """A 3DGS point cloud written byte by byte from the PLY format description (independent of
    save_ply), with properties deliberately out of numeric order to exercise the sort of
    load_model.py:22,31,37."""
    names = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] +
             [f"f_rest_{k}" for k in reversed(range(45))] + ["opacity", "scale_2", "scale_0", "scale_1"] +
             ["rot_0", "rot_1", "rot_2", "rot_3"])
    vals = rng.normal(size=(n, len(names))).astype(np.float32)
    with open(path, "wb") as f:
        f.write(b"ply\nformat " + fmt.encode() + b" 1.0\ncomment hand made\nelement vertex %d\n" % n)
        for nm in names:
            f.write(b"property float " + nm.encode() + b"\n")
        f.write(b"element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        if fmt == "ascii":
            for row in vals:
                f.write((" ".join(repr(float(v)) for v in row) + "\n").encode())
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            for row in vals:
                f.write(struct.pack(end + "%df" % len(names), *row))
    return {nm: vals[:, k] for k, nm in enumerate(names)}


@pytest.mark.parametrize("fmt", ["binary_little_endian", "binary_big_endian", "ascii"])
def test_ply_loader_field_mapping(tmp_path, fmt):
    import gsplat_amd as gs
    rng = np.random.default_rng(1)
    path = str(tmp_path / "point_cloud.ply")
    col = _handmade_ply(path, 17, rng, fmt)
    g = gs.load_data(path)
    assert g.center.shape == (17, 3) and g.spherical_harmonics.shape == (17, 3, 16) and g.degrees == 3
    assert g.opacity.shape == (17, 1) and g.scaling.shape == (17, 3) and g.rotation.shape == (17, 4)
    np.testing.assert_array_equal(g.center, np.stack([col["x"], col["y"], col["z"]], axis=1))
    np.testing.assert_array_equal(g.opacity[:, 0], col["opacity"])
    for c in range(3):
        np.testing.assert_array_equal(g.spherical_harmonics[:, c, 0], col[f"f_dc_{c}"])
        for k in range(15):      # load_model.py:27: reshape (P, 45) -> (P, 3, 15)
            np.testing.assert_array_equal(g.spherical_harmonics[:, c, 1 + k], col[f"f_rest_{c * 15 + k}"])
        np.testing.assert_array_equal(g.scaling[:, c], col[f"scale_{c}"])
    for k in range(4):
        np.testing.assert_array_equal(g.rotation[:, k], col[f"rot_{k}"])


def test_ply_and_npz_round_trip(tmp_path):
    import gsplat_amd as gs
    rng = np.random.default_rng(2)
    n = 33
    g = gs.GaussianSet(rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 3, 16)).astype(np.float32),
                       rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 4)).astype(np.float32),
                       rng.normal(size=(n, 1)).astype(np.float32))
    gs.save_ply(str(tmp_path / "a.ply"), g)
    h = gs.load_data(str(tmp_path / "a.ply"))
    for a in ("center", "spherical_harmonics", "scaling", "rotation", "opacity"):
        np.testing.assert_array_equal(np.asarray(getattr(h, a), dtype=np.float32), getattr(g, a))
    gs.save_gaussians(str(tmp_path / "iter_0_gaus"), g)
    with np.load(str(tmp_path / "iter_0_gaus.npz")) as d:         # gaussian.py:55-56 positional layout
        assert d.files == [f"arr_{k}" for k in range(6)]
        np.testing.assert_array_equal(d["arr_0"], g.center)
        np.testing.assert_array_equal(d["arr_1"], g.spherical_harmonics[:, :, 0])
        np.testing.assert_array_equal(d["arr_2"], g.spherical_harmonics)
    k = gs.load_gaussians(str(tmp_path / "iter_0_gaus"))
    for a in ("center", "spherical_harmonics", "scaling", "rotation", "opacity"):
        np.testing.assert_array_equal(getattr(k, a), getattr(g, a))


def test_ply_loader_errors(tmp_path):
    import gsplat_amd as gs
    p = tmp_path / "bad.ply"
    p.write_bytes(b"not a ply\n")
    with pytest.raises(ValueError):
        gs.load_data(str(p))
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nend_header\n\x00\x00")
    with pytest.raises(ValueError):
        gs.read_ply_vertices(str(p))
