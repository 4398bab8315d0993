"""gs_blend_backward_noalloc (include/gsplat.h): the allocation-free variant of gs_blend_backward, called
through the C ABI and compared with gs_blend_backward on the same forward state, two calls in a row, the
second with a smaller input (the static scratch keeps the first call's contents).

Bar: the two entry points run the same kernel on the same records; they differ only by the order of K7's
float atomics, so each gradient tensor must agree to max|a - b| <= 1e-4 * max|b| (north_star's fp32
tolerance).  The file collects last (zz): the round's end runs pytest -x."""
import ctypes

import numpy as np
import pytest

from conftest import synth_scene

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _both_backwards(ctx, N, W, H, seed, fx):
    import torch
    import gsplat_amd as gs
    from gsplat_amd import api
    from gsplat_amd._lib import check
    cam, R, t, g, rs = synth_scene(N, W, H, seed=seed, fx=fx)
    centers, depths, colors, conics, tiles_touched, radii = gs.forward_pass(
        ctx, None, None, cam, R, t, g, result_size=rs, tile_size=16, training=False)
    dev = ctx.device
    f32 = torch.float32
    cen = torch.from_numpy(np.float32(centers)).to(dev)
    col = torch.from_numpy(np.float32(colors)).to(dev)
    con = torch.from_numpy(np.float32(conics)).to(dev)
    opa = torch.from_numpy(np.float32(g.opacity)).to(dev)
    b = gs.match_gaus_to_tiles(ctx, None, None, tiles_touched, radii, depths, result_size=rs, want_keys=False)
    bg = api._bg3(np.array([0.2, 0.1, 0.3]))
    image, n_contrib, t_final = api.blend_forward(ctx, cen, col, opa, con, b, bg, 0.5)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    dL = torch.randn(W, H, 3, generator=gen).to(dev)
    ref = api.blend_backward(ctx, cen, col, opa, con, b, bg, 0.5, n_contrib, t_final, dL)
    out = (torch.full((N, 2), 7.0, dtype=f32, device=dev), torch.full((N, 3), 7.0, dtype=f32, device=dev),
           torch.full((N,), 7.0, dtype=f32, device=dev), torch.full((N, 3), 7.0, dtype=f32, device=dev))
    p = api._ptr
    check(ctx.lib.gs_blend_backward_noalloc(
        ctx.handle, N, p(cen), p(col), p(opa), p(con), p(b.vals), p(b.ranges), b.D, ctypes.byref(bg), W, H, 16, 0.5,
        p(n_contrib), p(t_final), p(dL), p(out[0]), p(out[1]), p(out[2]), p(out[3])), "gs_blend_backward_noalloc")
    torch.cuda.synchronize()
    return [r.cpu().numpy() for r in ref], [o.cpu().numpy() for o in out], b.D


def test_noalloc_backward_equals_backward_twice_second_smaller(gpu_ctx):
    for N, W, H, seed, fx in ((6000, 320, 240, 5, 200.0), (700, 96, 80, 6, 60.0)):
        ref, got, D = _both_backwards(gpu_ctx, N, W, H, seed, fx)
        assert D > N // 2
        for name, r, o in zip(("d_centers", "d_conics", "d_opacity", "d_colors"), ref, got):
            assert r.shape == o.shape and np.all(np.isfinite(o)), name
            scale = np.abs(r).max()
            assert scale > 0, name
            assert np.abs(o - r).max() <= TOL * scale, f"{name} N={N}: {np.abs(o - r).max():.3e} vs max {scale:.3e}"


def test_noalloc_rejects_sizes_beyond_the_static_scratch(gpu_ctx):
    import torch
    from gsplat_amd import api
    from gsplat_amd._lib import GsError, check
    ctx = gpu_ctx
    N = 65536 + 1
    dev = ctx.device
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    cen, col, opa, con = z(N, 2), z(N, 3), z(N), z(N, 3)
    ranges = torch.zeros((1, 2), dtype=torch.int32, device=dev)
    nc = torch.zeros((16, 16), dtype=torch.int32, device=dev)
    tf, dL = z(16, 16), z(16, 16, 3)
    bg = api._bg3(np.zeros(3))
    p = api._ptr
    with pytest.raises(GsError, match="exceeds the static scratch"):
        check(ctx.lib.gs_blend_backward_noalloc(
            ctx.handle, N, p(cen), p(col), p(opa), p(con), p(None), p(ranges), 0, ctypes.byref(bg), 16, 16, 16, 0.5,
            p(nc), p(tf), p(dL), p(cen), p(con), p(opa), p(col)), "gs_blend_backward_noalloc")
