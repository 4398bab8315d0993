"""CPU, world_size 2, gloo: the code that runs on the GPUs -- SplatPipeline + train_step with its lanes, the
early / late parameter-backward launches, the chunked K8 and the all-reduces -- driven through a context
whose `lib` is a Python stand-in for libgsplat_hip.so writing closed-form values through the same pointers
(CPU tensors).  Nothing in the package is replaced: only the object in the ctx slot differs, as the
reference's (ctx, queue, program) triple allows.  The expected gradients are computed independently in the
test, so a wrong pointer offset, a lost view or a missed / doubled reduction shows."""
import contextlib
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N, W, H = 1000, 8, 6          # N > 3 * 256: the chunked K8 runs over four ranges


def _f32(ptr, n):
    addr = ptr.value if hasattr(ptr, "value") else int(ptr)
    return np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr))


class FakeStream:
    def wait_stream(self, other):
        pass

    def wait_event(self, event):
        pass

    def record_event(self):
        return object()


class FakeLib:
    """The entry points train_step reaches, with closed-form outputs:
    image of view v = v + 1 everywhere; rows_v[g] = [(v+1)(g+1) + sum(dL_v), g, 0...];
    K8: d_means[g,:] (+)= sum_v rows_v[g,0]; d_quats[g,0] (+)= sum_v rows_v[g,1] (the gaussian's global
    index as the blend backward saw it); d_sh[g,:,:] (+)= n_views; d_opacity[g] (+)= 1."""

    def __init__(self):
        self.records = {}        # ctx handle -> address of the bound records
        self.log = []

    def gs_last_error(self):
        return b"fake"

    def gs_preprocess_batch_records(self, h, n, means, ls, quats, sh, opac, n_views, arr):
        for k in range(n_views):
            _f32(arr[k].records, n * 16)[0] = arr[k].cam.contents.fx       # the view's id travels in fx
        self.log.append(("k1", n_views))
        return 0

    def gs_ctx_use_records(self, h, n, rec):
        self.records[h] = getattr(rec, "value", None)
        return 0

    def gs_bin_count_culled_begin(self, h, *a):
        return 0

    def gs_bin_count_end(self, h, D):
        D._obj.value = 7
        return 0

    def gs_bin_fill_culled(self, h, *a):
        return 0

    def gs_bin_prepare_begin(self, h, n, n_views, views, *a):
        self.log.append(("prepare", n_views))
        return 0

    def gs_bin_prepare_end(self, h, n, n_views, views, w, hh, ts, D):
        for k in range(n_views):
            D[k] = 7
        return 0

    def gs_bin_fill_prepared(self, h, *a):
        return 0

    def gs_bin_bounds_batch(self, h, n, n_views, views, w, hh, ts, out):
        for k in range(n_views):
            out[2 * k], out[2 * k + 1] = 7, 1000
        return 0

    def gs_bin_culled_batch(self, h, n, n_views, views, w, hh, ts, off, depth_bits, capacity):
        assert capacity == 7 and depth_bits == 10
        self.log.append(("bin", n_views))
        return 0

    def _view(self, h):
        return float(_f32(self.records[h], 1)[0])

    def gs_blend_forward(self, h, n, cen, col, opa, con, vals, ranges, D, bg, w, hh, ts, off, image, nc, tf):
        _f32(image, w * hh * 3)[:] = self._view(h) + 1.0
        return 0

    def gs_image_loss(self, h, n, image, target, kind, scale, accumulate, dL, loss_out):
        d = _f32(image, n).astype(np.float64) - _f32(target, n)
        _f32(dL, n)[:] = (2.0 * scale / n) * d
        l = _f32(loss_out, 1)
        l[0] = (l[0] if accumulate else 0.0) + float((d * d).mean())
        return 0

    def gs_blend_forward_loss(self, h, n, cen, col, opa, con, vals, ranges, D, bg, w, hh, ts, off, image, nc, tf, target,
                              kind, scale, dL, loss_out):
        self.gs_blend_forward(h, n, cen, col, opa, con, vals, ranges, D, bg, w, hh, ts, off, image, nc, tf)
        return self.gs_image_loss(h, w * hh * 3, image, target, kind, scale, 1, dL, loss_out)

    def gs_blend_backward_rows(self, h, n, cen, col, opa, con, vals, ranges, D, bg, w, hh, ts, off, nc, tf, dL, rows):
        v = self._view(h)
        r = _f32(rows, n * 16).reshape(n, 16)
        r[:] = 0.0
        r[:, 0] = (v + 1.0) * (np.arange(n) + 1.0) + float(_f32(dL, w * hh * 3).astype(np.float64).sum())
        r[:, 1] = np.arange(n)
        return 0

    def gs_backward_params_batch(self, h, n, means, ls, quats, sh, opacity, is_logit, n_views, views, accumulate,
                                 d_means, d_ls, d_quats, d_opac, d_sh):
        outs = [_f32(d_means, n * 3).reshape(n, 3), _f32(d_ls, n * 3).reshape(n, 3), _f32(d_quats, n * 4).reshape(n, 4),
                _f32(d_opac, n), _f32(d_sh, n * 48).reshape(n, 48)] if n else []
        if not accumulate:
            for o in outs:
                o[:] = 0.0
        for k in range(n_views if n else 0):
            r = _f32(views[k].grad_rows, n * 16).reshape(n, 16)
            outs[0] += r[:, 0:1]
            outs[2][:, 0] += r[:, 1]
            outs[3] += 1.0
            outs[4] += 1.0
        self.log.append(("k8", n, n_views, int(accumulate)))
        return 0


class FakeCtx:
    """DeviceContext's interface on the CPU."""
    device = torch.device("cpu")
    device_index = 0
    _next = [1]

    def __init__(self, lib):
        self.lib = lib
        self.handle = self._next[0]
        self._next[0] += 1
        self._main = FakeStream()

    def sync_stream(self):
        pass

    def clone(self):
        return FakeCtx(self.lib)

    def new_stream(self, priority=0):
        return FakeStream()

    def current_stream(self):
        return self._main

    def use_stream(self, stream):
        return contextlib.nullcontext()

    def synchronize(self):
        pass


def _scene():
    import types
    rng = np.random.default_rng(0)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    return types.SimpleNamespace(center=f(N, 3), scaling=f(N, 3), rotation=f(N, 4), spherical_harmonics=f(N, 3, 16),
                                 opacity=f(N), degrees=3)


def _views_targets(gs, n_views):
    import types
    views = [gs.View(types.SimpleNamespace(params=[float(v), float(v)], width=W, height=H), np.eye(3), np.zeros(3))
             for v in range(n_views)]
    targets = [torch.full((W, H, 3), 0.25 * v, dtype=torch.float32) for v in range(n_views)]
    return views, targets


def _expected(n_views):
    """Closed form of what the whole batch must give."""
    n = W * H * 3
    g = np.arange(N, dtype=np.float64)
    means, loss = np.zeros(N), 0.0
    for v in range(n_views):
        d = (v + 1.0) - 0.25 * v
        loss += d * d
        means += (v + 1.0) * (g + 1.0) + np.float32(2.0 / n * d) * n
    return means, g * n_views, float(n_views), loss


def _check(grads, n_views, what):
    means, quat0, count, _ = _expected(n_views)
    np.testing.assert_allclose(grads.means.numpy(), np.repeat(means[:, None], 3, 1), rtol=2e-6, err_msg=what)
    np.testing.assert_allclose(grads.quats.numpy()[:, 0], quat0, rtol=1e-6, err_msg=what)       # offsets of every range
    assert np.all(grads.quats.numpy()[:, 1:] == 0) and np.all(grads.log_scales.numpy() == 0), what
    assert np.all(grads.sh.numpy() == count) and np.all(grads.opacity.numpy() == count), what


def _run(n_views, lanes, **step_args):
    import gsplat_amd as gs
    lib = FakeLib()
    pipe = gs.SplatPipeline(_scene(), [W, H], ctx=FakeCtx(lib))
    grads = gs.FlatGrads(N, torch.device("cpu"))
    grads.flat.fill_(123.0)                               # stale content must be overwritten, not added to
    views, targets = _views_targets(gs, n_views)
    loss, n_local = gs.train_step(pipe, views, targets, grads, concurrent_views=lanes, **step_args)
    return lib, grads, float(loss), n_local


@pytest.mark.parametrize("n_views,lanes", [(8, 4), (5, 4), (3, 2), (1, 4)])
def test_one_rank_real_step_code_matches_closed_form(n_views, lanes):
    lib, grads, loss, n_local = _run(n_views, lanes)
    assert n_local == n_views
    _check(grads, n_views, f"{n_views} views on {lanes} lanes")
    assert abs(loss - _expected(n_views)[3]) < 1e-4
    k8 = [e for e in lib.log if e[0] == "k8"]
    if n_views >= 4 and lanes == 4:      # early launch (write) for the first two lanes' views, late launch adds
        assert k8[0][3] == 0 and k8[-1][3] == 1 and k8[0][2] + k8[-1][2] == n_views
    assert sum(e[1] for e in lib.log if e[0] == "k1") == n_views


@pytest.mark.parametrize("n_views,lanes,group", [(8, 4, 4), (5, 3, 2), (1, 1, 4), (8, 2, 8)])
def test_one_rank_step_with_batched_lists_matches_closed_form(n_views, lanes, group):
    """train_step(batch_binning=True): lists per group of views (gs_bin_culled_batch), blends on side streams."""
    lib, grads, loss, n_local = _run(n_views, lanes, batch_binning=True, bin_group=group)
    _check(grads, n_views, f"{n_views} views, groups of {group}")
    assert abs(loss - _expected(n_views)[3]) < 1e-4
    assert sum(e[1] for e in lib.log if e[0] == "bin") == n_views
    assert max(e[1] for e in lib.log if e[0] == "bin") <= group
    assert sum(e[1] for e in lib.log if e[0] == "k1") == n_views


def _worker(rank, world, port, n_views, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    lib, grads, loss, n_local = _run(n_views, 4)
    torch.save({"flat": grads.flat.clone(), "n_local": n_local, "loss": loss, "log": lib.log},
               os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_views", [8, 5])
def test_two_ranks_hold_the_full_batch_gradient(tmp_path, n_views):
    import gsplat_amd as gs
    mp.spawn(_worker, args=(2, _free_port(), n_views, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt")) for k in range(2)]
    assert r[0]["n_local"] == len(gs.shard_views(n_views, 0, 2)) and r[1]["n_local"] == len(gs.shard_views(n_views, 1, 2))
    assert torch.equal(r[0]["flat"], r[1]["flat"])                   # replicas stay identical
    full = gs.FlatGrads(N, torch.device("cpu"))
    full.flat.copy_(r[0]["flat"])
    # view v went to rank v % 2: their ids differ, so the closed form over ALL views checks the sharding too
    _check(full, n_views, f"2 ranks, {n_views} views")
    assert abs(r[0]["loss"] + r[1]["loss"] - _expected(n_views)[3]) < 1e-4
    # the last parameter-backward launch ran over four ranges of gaussians on every rank
    for k in range(2):
        last = [e for e in r[k]["log"] if e[0] == "k8"][-4:]
        assert [e[1] for e in last] == [256, 256, 256, N - 768]


def test_flat_grads_views_alias_one_buffer():
    import gsplat_amd as gs
    g = gs.FlatGrads(5, torch.device("cpu"))
    offs, total = gs.flat_offsets(5)
    assert g.flat.numel() == total == 300 and offs == [0, 16, 32, 52, 60]     # groups start on 16-byte boundaries
    assert gs.flat_offsets(8) == ([0, 24, 48, 80, 88], 8 * 59)
    g.means.fill_(1); g.log_scales.fill_(2); g.quats.fill_(3); g.opacity.fill_(4); g.sh.fill_(5)
    assert g.flat.sum().item() == 5 * (3 * 1 + 3 * 2 + 4 * 3 + 4 + 48 * 5)
    g.zero_()
    assert g.flat.abs().sum().item() == 0


def test_gaussian_chunks_cover_the_range_on_aligned_starts():
    import gsplat_amd as gs
    for n, c in ((1000, 4), (1_000_000, 4), (255, 4), (0, 4), (10_000_000, 8), (1024, 1)):
        ch = gs.gaussian_chunks(n, c)
        assert ch[0][0] == 0 and ch[-1][1] == n and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        assert all(g0 % 256 == 0 for g0, _ in ch) and len(ch) <= max(c, 1)
