"""Round-3 robustness tests on the GPU, all through the C ABI: context options reach every lane, the batched
binning call guards the per-view state, the ordered K7 reduction is safe on capacity-sized lists, priority
streams are refused, the event wait is bounded, workspace accounting."""
import ctypes

import numpy as np
import pytest

from conftest import synth_scene

pytestmark = pytest.mark.gpu


def _views(gs, cam, R, n):
    return [gs.View(cam, R, np.array([0.05 * k, -0.02 * k, 0.1 * k])) for k in range(n)]


def test_deterministic_backward_holds_on_every_lane_of_a_step(gpu_ctx):
    """deterministic_backward set on the pipeline's ctx must reach the lane and alternate contexts (clones): two
    4-lane steps give bit-identical gradients, whether the option is set before the lanes exist or after."""
    import torch
    import gsplat_amd as gs
    W, H = 320, 240
    cam, R, t, g, rs = synth_scene(6000, W, H, seed=41, fx=200.0, log_scale=(1.5, 3.2))
    views = _views(gs, cam, R, 6)
    gen = torch.Generator(device="cpu").manual_seed(41)
    targets = [torch.rand(W, H, 3, generator=gen).to(gpu_ctx.device) for _ in views]

    def step(pipe):
        grads = gs.FlatGrads(pipe.N, gpu_ctx.device)
        gs.train_step(pipe, views, targets, grads, concurrent_views=4)
        torch.cuda.synchronize()
        return grads.flat.clone()

    # (a) lanes created first (atomics), option set afterwards: set_option must reach the existing clones
    late = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx)
    ref = step(late)
    gpu_ctx.set_option("deterministic_backward", 1)
    try:
        for c in late.contexts():
            assert c.options.get("deterministic_backward") == 1
        a, b = step(late), step(late)
        assert torch.equal(a, b)
        # (b) a pipeline whose lanes are cloned while the option is already on
        early = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx)
        c = step(early)
        assert all(x.options.get("deterministic_backward") == 1 for x in early.contexts())
        assert torch.equal(a, c)
    finally:
        gpu_ctx.set_option("deterministic_backward", 0)
    assert all(x.options.get("deterministic_backward") == 0 for x in late.contexts())
    scale = float(ref.abs().max())
    assert float((a - ref).abs().max()) <= 2e-5 * scale      # the atomics form agrees to rounding


def test_batched_binning_guards_the_per_view_state(gpu_ctx):
    """gs_bin_culled_batch overwrites the ctx's per-view workspace: it refuses to run over a pending
    gs_bin_count*_begin, and a gs_bin_fill* after it must not pass the state check of an older count."""
    import gsplat_amd as gs
    from bench import make_views
    cam, R, t, g, rs = synth_scene(4000, 256, 192, seed=5, log_scale=(1.0, 3.0))
    pipe = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx)
    views = make_views(gs, cam, 2)
    pres = pipe.preprocess_batch(views, records=True)
    tok = pipe.forward_begin(views[0], pre=pres[0])                 # a count is now pending on gpu_ctx
    with pytest.raises(gs.GsError, match="pending"):
        gs.api.bin_culled_batch(gpu_ctx, pres, rs, 16, pipe.pixel_offset)
    first = pipe.forward_bin(tok)                                   # the pending count is still answered correctly
    D = first.D
    batch = gs.api.bin_culled_batch(gpu_ctx, pres, rs, 16, pipe.pixel_offset)
    assert batch[0].true_D() == D
    # a fill with the OLD (N, D) must be refused now: the workspace holds the batch's numbers
    import torch
    vals = torch.empty(D, dtype=torch.int32, device=gpu_ctx.device)
    ranges = torch.empty((first.ranges.shape[0], 2), dtype=torch.int32, device=gpu_ctx.device)
    p = pres[0]
    with pipe_records(gs, gpu_ctx, pipe.N, p["rec"]):
        rc = gpu_ctx.lib.gs_bin_fill_culled(gpu_ctx.handle, pipe.N, gs.api._ptr(p["rect"]), gs.api._ptr(p["depths"]),
                                            None, None, None, None, rs[0], rs[1], 16, 0.5, D, None,
                                            gs.api._ptr(vals), gs.api._ptr(ranges))
    assert rc == -3          # GS_ERR_STATE


class pipe_records:
    def __init__(self, gs, c, N, rec):
        self.gs, self.c, self.N, self.rec = gs, c, N, rec

    def __enter__(self):
        assert self.c.lib.gs_ctx_use_records(self.c.handle, self.N, self.gs.api._ptr(self.rec)) == 0

    def __exit__(self, *exc):
        self.c.lib.gs_ctx_use_records(self.c.handle, 0, None)
        return False


def test_ordered_reduction_is_safe_on_capacity_sized_lists(gpu_ctx):
    """A list from gs_bin_culled_batch is handed to the blend calls with D = its capacity; the tail behind the last
    range is unwritten memory.  With deterministic_backward the entries are sorted by gaussian id: ids beyond N must
    never become a store address, and the gradients equal those of the per-view list bit for bit."""
    import torch
    import gsplat_amd as gs
    from bench import make_views
    W, H = 256, 192
    cam, R, t, g, rs = synth_scene(5000, W, H, seed=8, log_scale=(1.0, 3.0))
    pipe = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx)
    views = make_views(gs, cam, 1)
    pres = pipe.preprocess_batch(views, records=True)
    tok = pipe.forward_begin(views[0], pre=pres[0])
    ref = pipe.forward_bin(tok)
    cap = ref.D + 5000
    b = gs.api.bin_culled_batch(gpu_ctx, pres, rs, 16, pipe.pixel_offset, depth_bits=32, capacity=cap)[0]
    b.vals[ref.D:] = torch.randint(-2 ** 31, 2 ** 31 - 1, (cap - ref.D,), dtype=torch.int64,
                                   device=gpu_ctx.device).to(torch.int32)     # hostile tail
    dL = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(1)).to(gpu_ctx.device)
    guard = torch.full((4096,), 7.0, device=gpu_ctx.device)       # allocated right after the rows below
    gpu_ctx.set_option("deterministic_backward", 1)
    try:
        rows = {}
        for name, binning in (("ref", ref), ("cap", b)):
            _, st = pipe.forward_blend(dict(pre=pres[0], opac=tok["opac"], binning=binning, c=gpu_ctx))
            rows[name] = pipe.backward_rows(st, dL).clone()
    finally:
        gpu_ctx.set_option("deterministic_backward", 0)
    torch.cuda.synchronize()
    assert torch.equal(rows["ref"], rows["cap"])
    assert float(rows["ref"].abs().max()) > 0 and bool((guard == 7.0).all())


def test_priority_streams_are_refused(gpu_ctx):
    """The library launches on default-priority streams only (Concurrency): gs_ctx_set_stream reports an
    error for any other, and the context keeps its previous stream."""
    import torch
    import gsplat_amd as gs
    hi = torch.cuda.Stream(gpu_ctx.device, priority=-1)
    rc = gpu_ctx.lib.gs_ctx_set_stream(gpu_ctx.handle, ctypes.c_void_p(hi.cuda_stream))
    assert rc == -1 and b"priority" in gpu_ctx.lib.gs_last_error()
    with pytest.raises(gs.GsError, match="priority"):
        with gpu_ctx.use_stream(hi):
            gpu_ctx.sync_stream()
    gpu_ctx.sync_stream()
    cam, R, t, g, rs = synth_scene(500, 64, 48, seed=2, fx=60.0)
    image, _ = gs.SplatPipeline(g, rs, ctx=gpu_ctx).forward(gs.View(cam, R, t))
    assert bool(torch.isfinite(image).all())
    import inspect
    assert "priority" not in inspect.signature(gpu_ctx.new_stream).parameters


def test_count_wait_is_bounded(gpu_ctx):
    """gs_bin_count_end gives up with an error when the count does not complete in time (GSPLAT_WAIT_TIMEOUT_MS)
    instead of blocking for ever: a stream held back by 0.3 s of queued work stands in for a stuck one."""
    import os
    import torch
    import gsplat_amd as gs
    os.environ["GSPLAT_WAIT_TIMEOUT_MS"] = "20"
    try:
        c = gs.DeviceContext(gpu_ctx.device_index)
    finally:
        del os.environ["GSPLAT_WAIT_TIMEOUT_MS"]
    cam, R, t, g, rs = synth_scene(2000, 128, 96, seed=3, fx=100.0)
    pipe = gs.SplatPipeline(g, rs, ctx=c)
    s = c.new_stream()
    with c.use_stream(s):
        pipe.forward(gs.View(cam, R, t))               # sizes the workspace (a first call allocates, which synchronises)
        x = torch.zeros(2 ** 28, dtype=torch.float32, device=c.device)
        for _ in range(600):                           # ~0.3 s of device time in front of the count
            x.add_(1.0)
        tok = pipe.forward_begin(gs.View(cam, R, t))
        with pytest.raises(gs.GsError, match="has not finished"):
            pipe.forward_bin(tok)
    torch.cuda.synchronize()
    image, _ = pipe.forward(gs.View(cam, R, t))         # the context is usable afterwards
    assert bool(torch.isfinite(image).all())
    c.close()


def test_workspace_accounting(gpu_ctx):
    import gsplat_amd as gs
    c = gs.DeviceContext(gpu_ctx.device_index)
    assert c.info("workspace_bytes") == 0 and c.info("allocations") == 0
    cam, R, t, g, rs = synth_scene(3000, 128, 96, seed=3, fx=100.0)
    pipe = gs.SplatPipeline(g, rs, ctx=c)
    pipe.forward(gs.View(cam, R, t))
    a1, b1 = c.info("allocations"), c.info("workspace_bytes")
    assert a1 > 0 and b1 > 0
    pipe.forward(gs.View(cam, R, t))
    assert c.info("allocations") == a1 and c.info("workspace_bytes") == b1     # steady state: no hipMalloc
    with pytest.raises(gs.GsError):
        c.info("nonsense")
    c.close()


@pytest.mark.parametrize("culled", [False, True])
@pytest.mark.parametrize("W,H,tile", [(640, 360, 16), (203, 117, 32)])
def test_prepared_batch_lists_equal_the_per_view_calls(gpu_ctx, culled, W, H, tile):
    """gs_bin_prepare_begin / _end + gs_bin_fill_prepared (count, depth order and offsets per batch of views; emit,
    tile sort and ranges per view, here on another ctx) against gs_bin_count[_culled] + gs_bin_fill[_culled] view by
    view: D, keys, values and ranges bit for bit, for the reference's lists and the culled ones."""
    import torch
    import gsplat_amd as gs
    from bench import make_views
    cam, R, t, g, rs = synth_scene(50_000, W, H, seed=23, log_scale=(0.5, 3.0))
    pipe = gs.SplatPipeline(g, rs, tile_size=tile, opacity_is_logit=False, ctx=gpu_ctx, cull_tiles=culled)
    views = make_views(gs, cam, 5)
    pres = pipe.preprocess_batch(views)
    other = gpu_ctx.clone()
    tok = gs.api.bin_prepare_begin(other, pres, rs, tile, pipe.pixel_offset, culled)
    with pytest.raises(gs.GsError, match="pending"):
        gs.api.bin_prepare_begin(other, pres, rs, tile, pipe.pixel_offset, culled)     # one batch per ctx
    Ds = gs.api.bin_prepare_end(tok)
    for k, p in enumerate(pres):
        ref = pipe.forward_bin(pipe.forward_begin(views[k], pre=p), want_keys=True)
        with pipe_records(gs, gpu_ctx, pipe.N, p["rec"]):
            got = gs.api.bin_fill_prepared(gpu_ctx, p, want_keys=True)
        assert Ds[k] == got.D == ref.D > 0
        assert torch.equal(got.ranges, ref.ranges) and torch.equal(got.vals, ref.vals) and torch.equal(got.keys, ref.keys)
    other.close()


@pytest.mark.parametrize("kind", ["l1", "l2"])
@pytest.mark.parametrize("W,H", [(320, 240), (203, 117)])
def test_blend_forward_with_the_loss_in_its_epilogue(gpu_ctx, kind, W, H):
    """gs_blend_forward_loss against gs_blend_forward followed by gs_image_loss: image, n_contrib, t_final and
    dL/dimage bit for bit (the same arithmetic per pixel), the loss to float-atomics rounding, added to loss_out; and
    train_step with and without the fused form gives the same gradients."""
    import torch
    import gsplat_amd as gs
    cam, R, t, g, rs = synth_scene(8000, W, H, seed=9, fx=0.6 * W, log_scale=(1.5, 3.2))
    pipe = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx, background=(.1, .2, .3))
    view = gs.View(cam, R, t)
    target = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(3)).to(gpu_ctx.device)
    img0, st0 = pipe.forward(view)
    dL0, loss0 = pipe.image_loss(img0, target, kind, 0.7)
    acc = torch.full((1,), 2.5, dtype=torch.float32, device=gpu_ctx.device)
    img1, st1 = pipe.forward(view, loss=(target, kind, 0.7, acc))
    torch.cuda.synchronize()
    assert torch.equal(img0, img1) and torch.equal(st0["n_contrib"], st1["n_contrib"]) and torch.equal(st0["t_final"], st1["t_final"])
    assert torch.equal(dL0, st1["dL"])
    assert abs(float(acc) - 2.5 - float(loss0)) <= 2e-6 * max(float(loss0), 1.0) + 3e-7
    views = [view, gs.View(cam, R, np.array([0.1, -0.05, 0.2])), gs.View(cam, R, np.array([-0.1, 0.05, 0.1]))]
    targets = [target] * 3
    ga, gb = gs.FlatGrads(pipe.N, gpu_ctx.device), gs.FlatGrads(pipe.N, gpu_ctx.device)
    la, _ = gs.train_step(pipe, views, targets, ga, concurrent_views=2, loss_kind=kind, fused_loss=True)
    lb, _ = gs.train_step(pipe, views, targets, gb, concurrent_views=2, loss_kind=kind, fused_loss=False)
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 3e-6 * float(lb)
    assert float((ga.flat - gb.flat).abs().max()) <= 2e-5 * float(gb.flat.abs().max())


def test_k1_fast_form_equals_the_general_form(gpu_ctx, monkeypatch):
    """K1's batched kernel has a form for launches whose views all have SH degree 3, the strict clip-space centre and
    a power-of-two tile (constant-folded degree tests, tile divisions as exact multiplications) and a general one;
    GS_K1_GENERAL=1 forces the general one.  Every output of both batched entry points is bit-identical between
    the two, near-culled, off-screen and clamped-colour gaussians included; a tile size of 24 takes the general
    form by itself and must agree with the single-view kernel."""
    import torch
    import gsplat_amd as gs
    cam, R, t, g, rs = synth_scene(40_000, 640, 360, seed=77, fx=300.0, log_scale=(0.5, 3.5))
    views = [gs.View(cam, R, np.array([0.3 * k, -0.2 * k, 0.6 * k - 4.0])) for k in range(5)]
    pipe = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx)
    out = {}
    for form in ("fast", "general"):
        if form == "general":
            monkeypatch.setenv("GS_K1_GENERAL", "1")
        rec = pipe.preprocess_batch(views, records=True)
        arr = pipe.preprocess_batch(views, records=False)
        torch.cuda.synchronize()
        out[form] = ([{k: p[k].clone() for k in ("rec", "depths", "clamped", "rect", "radii")} for p in rec],
                     [{k: p[k].clone() for k in ("centers", "colors", "conics", "depths", "clamped", "rect", "radii")}
                      for p in arr])
    monkeypatch.delenv("GS_K1_GENERAL")
    for a, b in zip(out["fast"][0] + out["fast"][1], out["general"][0] + out["general"][1]):
        for k in a:
            assert torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8)), k
    assert int(out["fast"][0][0]["radii"].count_nonzero()) > 1000
    assert int(out["fast"][0][0]["clamped"].count_nonzero()) > 0
    assert int((out["fast"][0][0]["radii"] == 0).sum()) > 0
    # a tile size that is not a power of two: the launch takes the general form on its own
    pipe24 = gs.SplatPipeline(g, rs, tile_size=24, opacity_is_logit=False, ctx=gpu_ctx)
    from gsplat_amd import api
    pre = pipe24.preprocess_batch(views[:2], records=False)
    null = ctypes.c_void_p(0)
    for v, p in zip(views[:2], pre):
        one = pipe24._alloc_pre(api.make_camera(v.camera, v.camera_r, v.camera_t, pipe24.result_size, 24,
                                                pipe24.degrees, pipe24.cam_flags))
        api.check(gpu_ctx.lib.gs_preprocess(
            gpu_ctx.handle, pipe24.N, api._ptr(pipe24.means), api._ptr(pipe24.log_scales), api._ptr(pipe24.quats),
            api._ptr(pipe24.sh), ctypes.byref(one["cam"]), api._ptr(one["centers"]), api._ptr(one["depths"]),
            api._ptr(one["colors"]), api._ptr(one["conics"]), api._ptr(one["clamped"]), api._ptr(one["rect"]),
            api._ptr(one["radii"]), null, null, null, null, null, null), "gs_preprocess")
        torch.cuda.synchronize()
        for k in ("centers", "depths", "colors", "conics", "clamped", "rect", "radii"):
            assert torch.equal(one[k], p[k]), k


def test_k8_fast_form_equals_the_general_form(gpu_ctx, monkeypatch):
    """K8 has the same two forms as K1 (SH degree 3 and the strict clip-space centre folded in at compile time, or
    read from the camera); GS_K8_GENERAL=1 forces the general one.  Same gradient rows in, the same parameter
    gradients out to fp32 rounding (the file is compiled with FMA contraction and the two forms contract differently),
    for the write and the accumulate call."""
    import torch
    import gsplat_amd as gs
    W, H = 320, 240
    cam, R, t, g, rs = synth_scene(20_000, W, H, seed=78, fx=200.0, log_scale=(1.0, 3.2))
    pipe = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx)
    gen = torch.Generator(device="cpu").manual_seed(78)
    states = []
    for v in _views(gs, cam, R, 3):
        _, st = pipe.forward(v)
        pipe.backward_rows(st, torch.randn(W, H, 3, generator=gen).to(gpu_ctx.device))
        states.append(st)
    out = {}
    for form in ("fast", "general"):
        if form == "general":
            monkeypatch.setenv("GS_K8_GENERAL", "1")
        grads = gs.FlatGrads(pipe.N, gpu_ctx.device)
        pipe.backward_params(states, grads, accumulate=False)
        pipe.backward_params(states[:2], grads, accumulate=True)
        torch.cuda.synchronize()
        out[form] = grads.flat.clone()
    monkeypatch.delenv("GS_K8_GENERAL")
    scale = float(out["fast"].abs().max())
    assert scale > 0
    assert float((out["fast"] - out["general"]).abs().max()) <= 2e-6 * scale


def test_sorted_count_form_gives_the_same_lists(gpu_ctx):
    """From `sorted_count_from` gaussians on the depth sort runs before the tile count and count / scan / emit stream
    depth-ordered arrays (large scenes); sorted_count_from = 0 forces that form at any size.  Keys, values and ranges
    are bit-identical to the default form: random rects incl. degenerate and off-grid ones, zero / tied / wrapped
    (negative) depths, the 64-bit keys, and the culled lists of a synthetic scene through the pipeline."""
    import torch
    import gsplat_amd as gs
    rng = np.random.default_rng(77)
    cases = []
    for case in range(16):
        N = int(rng.choice([1, 63, 64, 65, 1000, 4097, 20000]))
        ts = int(rng.choice([8, 16, 32]))
        W, H = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        tx_n, ty_n = -(-W // ts), -(-H // ts)
        x0 = rng.integers(-2, tx_n + 2, N); y0 = rng.integers(-2, ty_n + 2, N)
        x1 = x0 + rng.integers(-1, 6, N); y1 = y0 + rng.integers(-1, 6, N)
        tt = np.stack([np.stack([x0, y0], 1), np.stack([x1, y1], 1)], 1).astype(np.int32)
        dep = [rng.integers(0, 40, N), rng.integers(2000, 60000, N), rng.integers(0, 2**24, N),
               rng.integers(-2**31, 2**31, N)][case % 4].astype(np.int32)
        cases.append((tt, dep, [W, H], ts))
    cam, R, t, g, rs = synth_scene(30_000, 640, 360, seed=79, fx=300.0, log_scale=(1.0, 3.3))
    views = _views(gs, cam, R, 2)

    def run():
        out = []
        for tt, dep, rs_, ts in cases:
            b = gs.match_gaus_to_tiles(gpu_ctx, None, None, tt, np.ones(len(dep)), dep, result_size=rs_, tile_size=ts)
            out.append((b.D, b.keys.cpu().numpy().copy() if b.D else None, b.vals.cpu().numpy().copy() if b.D else None,
                        b.ranges.cpu().numpy().copy()))
        for cull in (False, True):
            pipe = gs.SplatPipeline(g, rs, opacity_is_logit=False, ctx=gpu_ctx, cull_tiles=cull)
            for v in views:
                image, st = pipe.forward(v)
                torch.cuda.synchronize()
                b = st["binning"]
                out.append((b.D, None, b.vals.cpu().numpy().copy(), b.ranges.cpu().numpy().copy(), image.cpu().numpy().copy()))
        return out

    ref = run()
    gpu_ctx.set_option("sorted_count_from", 0)
    try:
        got = run()
    finally:
        gpu_ctx.set_option("sorted_count_from", 2 << 20)
    # the culled passes walk the ellipse's own bounding box instead of the whole rect (cull_shrink_rect, default on): it
    # only skips pairs the exact test rejects, so the lists are the same without it, in either count form
    gpu_ctx.set_option("cull_shrink_rect", 0)
    try:
        plain = run()
        gpu_ctx.set_option("sorted_count_from", 0)
        plain_sorted = run()
    finally:
        gpu_ctx.set_option("cull_shrink_rect", 1)
        gpu_ctx.set_option("sorted_count_from", 2 << 20)
    for other in (got, plain, plain_sorted):
        assert len(ref) == len(other)
        for k, (a, b) in enumerate(zip(ref, other)):
            assert a[0] == b[0], k
            for x, y in zip(a[1:], b[1:]):
                assert (x is None and y is None) or np.array_equal(x, y), k
    assert sum(a[0] for a in ref) > 100_000


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 66, 67, 130])
@pytest.mark.parametrize("det", [False, True])
def test_k7_row_sums_for_every_remainder_of_parked_entries(gpu_ctx, n, det):
    """K7 forms the 16-lane row sums of four entries at a time through LDS (csrc/blend.hip, sum_rows) and finishes a
    batch with one to three parked entries: n faint gaussians that all cover the whole 48x32 frame (nothing
    terminates, every entry is reduced in every tile) make each tile's list exactly n entries long, so the last
    round of every 64-entry batch holds n mod 4 (and (n - 64) mod 4, ...) entries.  All nine outputs against fp64
    autograd of the oracle blend told the device's decisions, 1e-4 of the largest entry; with and without the
    ordered (deterministic) reduction."""
    import torch
    import gsplat_amd as gs
    from oracle import ref_np, ref_torch
    W, H = 48, 32
    rng = np.random.default_rng(400 + n)
    cen = np.stack([rng.uniform(4, W - 4, n), rng.uniform(4, H - 4, n)], 1).astype(np.float32)
    s = rng.uniform(25.0, 60.0, (n, 2))
    th = rng.uniform(0, np.pi, n)
    con = np.zeros((n, 3), np.float32)
    for i in range(n):
        Rm = np.array([[np.cos(th[i]), -np.sin(th[i])], [np.sin(th[i]), np.cos(th[i])]])
        inv = np.linalg.inv(Rm @ np.diag(s[i] ** 2) @ Rm.T)
        con[i] = inv[0, 0], inv[0, 1], inv[1, 1]
    opa = rng.uniform(0.02, 0.05 if n > 8 else 0.4, n).astype(np.float32)     # 130 of them still leave T > 1e-4
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    rs = [W, H]
    tt = np.tile(np.array([[[0, 0], [W // 16, H // 16]]], dtype=np.int32), (n, 1, 1))
    dep = rng.permutation(n).astype(np.int32) + 1
    dev = gpu_ctx.device
    t_ = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    gpu_ctx.set_option("deterministic_backward", 1 if det else 0)
    try:
        b = gs.match_gaus_to_tiles(gpu_ctx, None, None, tt, np.ones(n), dep, result_size=rs, tile_size=16)
        bg = np.array([0.3, 0.1, 0.2])
        bg3 = gs.api._bg3(bg)
        img, nc, tf = gs.api.blend_forward(gpu_ctx, t_(cen), t_(col), t_(opa), t_(con), b, bg3, 0.5)
        vals, ranges = b.vals.cpu().numpy(), b.ranges.cpu().numpy()
        assert np.all(ranges[:, 1] - ranges[:, 0] == n)
        assert int((nc.cpu().numpy() & 0x7fffffff).min()) == n            # every pixel walked the whole list
        forced = ref_np.device_decisions(cen, con, opa, vals, ranges, rs, 16, 0.5, nc.cpu().numpy())
        p = [torch.tensor(a.astype(np.float64), requires_grad=True) for a in (cen, col, opa, con)]
        ref_img = ref_torch.blend(p[0], p[1], p[2], p[3], vals, ranges, rs, 16, bg, 0.5, forced=forced)
        wimg = np.random.default_rng(9).normal(size=(W, H, 3)).astype(np.float32)
        (ref_img * torch.tensor(wimg, dtype=torch.float64)).sum().backward()
        got = gs.api.blend_backward(gpu_ctx, t_(cen), t_(col), t_(opa), t_(con), b, bg3, 0.5, nc, tf, t_(wimg))
        for g_, want, what in zip(got, (p[0].grad, p[3].grad, p[2].grad, p[1].grad), ("centers", "conics", "opacity", "colors")):
            gnp = g_.cpu().numpy().astype(np.float64)
            ref = want.numpy()
            assert np.all(np.isfinite(gnp)), what
            scale = np.abs(ref).max()
            assert scale > 0 and np.abs(gnp - ref).max() <= 1e-4 * scale, (what, n, np.abs(gnp - ref).max() / scale)
    finally:
        gpu_ctx.set_option("deterministic_backward", 0)
