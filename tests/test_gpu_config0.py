"""BASELINE.json configs[0] end to end on the GPU box, through the reference-shaped API: a ~10 k point COLMAP scene
written by the reference's own write_model (tests/golden/colmap_scene, oracle/make_colmap_scene_fixture.py) goes
through read_model -> gaussians_from_point_cloud (gaussian.py:9-29) -> forward_pass -> match_gaus_to_tiles ->
c_rasterize (the rasterize.c path, rasterization.py:109-153) and gpu_rasterize at 256x256 / 16x16 tiles
(main.py:9-23, train.py:13,28-37).  Pinned by what the reference's own code computed from the same files
(expected.npz: kNN distances, forward_pass outputs of three images, a pure-Python rasterize() crop) and by the C
restatement of rasterize.c for the full frame (image and last_contributors)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import cref, ref_np

pytestmark = pytest.mark.gpu
SCENE = os.path.join(GOLDEN, "colmap_scene")
RS = [256, 256]
IMG_ATOL, IMG_RTOL = 2e-5, 1e-4


@pytest.fixture(scope="module")
def scene():
    import gsplat_amd as gs
    exp = np.load(os.path.join(SCENE, "expected.npz"), allow_pickle=False)
    cameras, images, points = gs.read_model(SCENE, ".bin")
    assert len(points) == int(exp["n_points"]) == 10_000 and sorted(images) == [1, 2, 3]
    g = gs.gaussians_from_point_cloud(points)
    np.testing.assert_allclose(np.exp(g.scaling[:, 0]), np.clip(exp["knn_distances"], 1e-7, None), rtol=1e-12)
    return exp, cameras, images, g


def _forward(gs, ctx, cameras, im, g):
    R = gs.qvec2rotmat(im.qvec)                      # main.py / train.py:25-26 build R from the image's quaternion
    return R, gs.forward_pass(ctx, None, None, cameras[im.camera_id], R, np.asarray(im.tvec), g, result_size=RS,
                              tile_size=16, training=False)


@pytest.mark.parametrize("iid", [1, 2, 3])
def test_forward_pass_and_lists_match_the_reference(gpu_ctx, scene, iid):
    import gsplat_amd as gs
    exp, cameras, images, g = scene
    R, (centers, depths, colors, conics, tiles_touched, radii) = _forward(gs, gpu_ctx, cameras, images[iid], g)
    np.testing.assert_allclose(R, exp[f"img{iid}_R"], rtol=0, atol=1e-15)
    assert np.array_equal(depths, exp[f"img{iid}_depths"])                       # bit-exact integers
    assert np.array_equal(tiles_touched, exp[f"img{iid}_tiles_touched"])
    assert np.array_equal(radii, exp[f"img{iid}_radii"])
    np.testing.assert_allclose(centers[::8], exp[f"img{iid}_centers"], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(conics[::8], exp[f"img{iid}_conics"], rtol=4e-6, atol=1e-7)
    np.testing.assert_allclose(colors[::8], exp[f"img{iid}_colors"], rtol=2e-6, atol=1e-6)
    b = gs.match_gaus_to_tiles(gpu_ctx, None, None, tiles_touched, radii, depths, result_size=RS, tile_size=16)
    keys, vals, ranges = ref_np.bin_duplicates(exp[f"img{iid}_tiles_touched"], exp[f"img{iid}_depths"], RS, 16)
    assert b.D == len(keys) > 15_000
    assert np.array_equal(b.keys.cpu().numpy().view(np.uint64), keys)
    assert np.array_equal(b.vals.cpu().numpy(), vals) and np.array_equal(b.ranges.cpu().numpy(), ranges)


@pytest.mark.parametrize("iid", [1, 3])
def test_c_rasterize_and_gpu_rasterize_match_the_c_oracle(gpu_ctx, scene, iid):
    """The full 256x256 frame through rasterize_tile (per tile, host arrays: the rasterize.c path of configs[0]) and
    through gpu_rasterize, against the restatement of rasterize.c (pixel offset +0.5): images within tolerance,
    last_contributors bit-exact."""
    import gsplat_amd as gs
    exp, cameras, images, g = scene
    _, (centers, depths, colors, conics, tiles_touched, radii) = _forward(gs, gpu_ctx, cameras, images[iid], g)
    b = gs.match_gaus_to_tiles(gpu_ctx, None, None, tiles_touched, radii, depths, result_size=RS, tile_size=16)
    vals, ranges = b.vals.cpu().numpy(), b.ranges.cpu().numpy()
    opac = np.float32(g.opacity)                                                  # raw opacities, train.py:37
    ref, lc, tf, fragile, _ = cref.rasterize_frame(centers, colors, opac, conics, vals, ranges, RS, (0, 0, 0),
                                                   pixel_offset=0.5, margin=1e-4)
    ok = ~fragile.astype(bool)
    assert fragile.mean() < 5e-3 and ref.max() > 0.2

    def check(img, what):
        err = np.abs(img - ref)
        assert np.all(err[ok] <= IMG_ATOL + IMG_RTOL * np.abs(ref)[ok]), f"{what}: max err {err[ok].max():.3e}"
        assert err.max() < 0.02, what

    img_c = gs.c_rasterize(centers, colors, opac, conics, b, training=False, result_size=RS, tile_size=16,
                           background_color=np.zeros(3))
    check(img_c, "c_rasterize")
    assert np.array_equal(gs.c_rasterize.last_contributors[ok], lc[ok])
    img_g, n_contrib, t_final = gs.gpu_rasterize(gpu_ctx, None, None, centers, colors, opac, conics, b, training=False,
                                                 result_size=RS, tile_size=16, return_state=True)
    check(img_g, "gpu_rasterize")
    assert np.array_equal((n_contrib.cpu().numpy() & 0x7FFFFFFF)[ok], lc[ok])
    np.testing.assert_allclose(t_final.cpu().numpy()[ok], tf[ok], rtol=1e-4, atol=1e-7)
    # the two entry points run the same kernel: identical pixels
    assert np.array_equal(np.float32(img_c), np.float32(img_g))


def test_rasterize_matches_the_reference_python_blend_on_a_crop(gpu_ctx, scene):
    """rasterize() (rasterization.py:155-215, pixel centre c - (x + .5)) against the image the reference's own
    interpreted loop produced for tiles [2,8) x [5,9) of image 1, given the same depth-keyed dicts."""
    import gsplat_amd as gs
    exp, cameras, images, g = scene
    _, (centers, depths, colors, conics, tiles_touched, radii) = _forward(gs, gpu_ctx, cameras, images[1], g)
    x0, x1, y0, y1 = (int(v) for v in exp["img1_crop_tiles"])
    mk = [[dict() for _ in range(y1 - y0)] for _ in range(x1 - x0)]
    for i in np.nonzero(depths)[0]:
        for tx in range(max(tiles_touched[i, 0, 0], x0), min(tiles_touched[i, 1, 0], x1)):
            for ty in range(max(tiles_touched[i, 0, 1], y0), min(tiles_touched[i, 1, 1], y1)):
                mk[tx - x0][ty - y0][int(depths[i])] = int(i)
    shifted = centers - np.array([x0 * 16.0, y0 * 16.0])
    rs = [(x1 - x0) * 16, (y1 - y0) * 16]
    img = gs.rasterize(shifted, colors, g.opacity, conics, mk, training=False, result_size=rs, tile_size=16,
                       background_color=np.zeros(3), ctx=gpu_ctx)
    ref = exp["img1_crop_image"]
    assert img.shape == ref.shape == (96, 64, 3)
    b = gs.api._binning_from_lists(mk, rs, 16, gpu_ctx.device)
    _, _, _, fragile, _ = cref.rasterize_frame(shifted, colors, np.float32(g.opacity), conics, b.vals.cpu().numpy(),
                                               b.ranges.cpu().numpy(), rs, (0, 0, 0), pixel_offset=-0.5, margin=1e-4)
    ok = ~fragile.astype(bool)
    err = np.abs(img - ref)
    assert fragile.mean() < 5e-3 and np.all(err[ok] <= IMG_ATOL + IMG_RTOL * np.abs(ref)[ok]) and err.max() < 0.02
