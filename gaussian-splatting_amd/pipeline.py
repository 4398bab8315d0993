"""Device-resident chaining of the hot path for rendering and multi-view training steps.

One view = K1 preprocess -> K2-K5 binning -> K6 blend forward [-> K7 blend backward -> K8
parameter backward].  Nothing leaves the device between stages (the reference re-uploads every
array on every call, rasterization.py:44-55, backward_pass.py:106-120).  A training step renders a
batch of views; with torch.distributed initialised the views are sharded round-robin over the ranks
(one process per GPU) and the flat per-gaussian gradient buffer [N*59] is summed with ONE
all-reduce (RCCL over xGMI when the backend is "nccl").  The reference has no counterpart for the
batch/all-reduce part (its loop is one view per iteration, train.py:18-43).
"""
import ctypes

import numpy as np
import torch

from . import api
from ._lib import GsViewRecords, GsViewBackward, GsViewPreprocess, check

GRAD_FLOATS = 59   # 3 mean + 3 log-scale + 4 quaternion + 1 opacity + 48 SH


GROUP_WIDTHS = (3, 3, 4, 1, 48)   # means, log_scales, quats, opacity, sh


def flat_offsets(N):
    """Float offsets of the five parameter groups inside the flat gradient / moment buffers and the
    total length.  Every group starts at a multiple of 4 floats (16 bytes: the kernels store
    float4s), so for N % 4 != 0 the buffer is a few floats longer than N*59."""
    offs, o = [], 0
    for w in GROUP_WIDTHS:
        offs.append(o)
        o = (o + w * N + 3) // 4 * 4
    return offs, o


class FlatGrads:
    """One contiguous fp32 buffer (N*59 floats, plus alignment padding) with per-parameter views,
    so a single collective moves every gradient."""

    def __init__(self, N, device):
        self.N = N
        offs, total = flat_offsets(N)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.means = self.flat[offs[0]:offs[0] + 3 * N].view(N, 3)
        self.log_scales = self.flat[offs[1]:offs[1] + 3 * N].view(N, 3)
        self.quats = self.flat[offs[2]:offs[2] + 4 * N].view(N, 4)
        self.opacity = self.flat[offs[3]:offs[3] + N]
        self.sh = self.flat[offs[4]:offs[4] + 48 * N].view(N, 3, 16)

    def zero_(self):
        self.flat.zero_()


class _records:
    """Binds caller-owned blend records to a ctx for the calls inside the block (gs_ctx_use_records) and
    unbinds them on the way out, also when one of those calls raises."""

    def __init__(self, c, N, rec):
        self.c, self.N, self.rec = c, N, rec

    def __enter__(self):
        check(self.c.lib.gs_ctx_use_records(self.c.handle, self.N if self.rec is not None else 0, api._ptr(self.rec)),
              "gs_ctx_use_records")

    def __exit__(self, *exc):
        if self.rec is not None:
            self.c.lib.gs_ctx_use_records(self.c.handle, 0, None)
        return False


class View:
    def __init__(self, camera, camera_r, camera_t):
        self.camera, self.camera_r, self.camera_t = camera, camera_r, camera_t


class SplatPipeline:
    """Holds the parameters on one device and runs views through the native kernels."""

    def __init__(self, gaussians, result_size, tile_size=16, background=(0., 0., 0.), pixel_offset=0.5,
                 opacity_is_logit=True, ctx=None, cull_tiles=True, apply_view=False, diag_scale=False):
        self.ctx = api._ctx(ctx)
        self.cam_flags = api.camera_flags(apply_view, diag_scale)   # 0 = the reference's quirks Q1/Q4 kept
        self.cull_tiles = bool(cull_tiles)   # gs_bin_*_culled: same images/gradients, about half the list
        dev = self.ctx.device
        self.N, self.means, self.log_scales, self.quats, self.sh = api._gaussian_tensors(gaussians, dev)
        self.opacity = api._dev(gaussians.opacity, torch.float32, dev, (self.N,))
        self.degrees = int(gaussians.degrees)
        self.result_size = [int(result_size[0]), int(result_size[1])]
        self.tile_size = int(tile_size)
        self.bg = api._bg3(np.asarray(background, dtype=np.float64))
        self.pixel_offset = float(pixel_offset)
        self.opacity_is_logit = bool(opacity_is_logit)
        self.stats = {}
        self._lanes = []
        self._alts = []

    def lanes(self, n):
        """n (DeviceContext, torch stream) pairs for rendering independent views concurrently: each
        lane owns its workspace arena and stream, so one view's blend tail overlaps another view's
        preprocess/binning.  The parameters are shared (read-only during a step)."""
        while len(self._lanes) < n:
            self._lanes.append((self.ctx.clone(), self.ctx.new_stream()))
        return self._lanes[:n]

    def alt_contexts(self, n):
        """A second DeviceContext (workspace) per lane, main stream included: with two, a lane can
        count the tiles of its next view while the current view's records are still in use."""
        while len(self._alts) < n:
            self._alts.append(self.ctx.clone())
        return self._alts[:n]

    def contexts(self):
        """Every DeviceContext this pipeline launches on (for profiling)."""
        return [self.ctx] + [c for c, _ in self._lanes] + list(self._alts)

    def reserve(self, n_max, d_max, concurrent_views=4):
        """Reserve ahead on every context a step with `concurrent_views` lanes launches on (main, lanes, their
        alternates): workspace for views of up to n_max gaussians and d_max (gaussian, tile) pairs.  A training loop
        whose gaussian count grows (densification) calls this once with its budget; afterwards
        sum(c.info("allocations") for c in contexts()) stays constant."""
        n = max(1, int(concurrent_views))
        self.lanes(n - 1 if n > 3 else n)
        self.alt_contexts(n)
        for c in self.contexts():
            # the first two alternates hold the batches of gs_bin_prepare_* that a step keeps in flight
            batch = 8 if any(c is a for a in self._alts[:2]) else 0
            c.reserve(n_max, d_max, self.result_size, self.tile_size, batch_views=batch)

    def workspace_bytes(self):
        return sum(c.info("workspace_bytes") for c in self.contexts())

    def allocations(self):
        return sum(c.info("allocations") for c in self.contexts())

    def export(self):
        """The current parameters as a host GaussianSet (float64 NumPy, the reference's dtypes)."""
        f = lambda t: t.detach().cpu().numpy().astype(np.float64)
        return api.GaussianSet(f(self.means), f(self.sh), f(self.log_scales), f(self.quats), f(self.opacity),
                               self.degrees)

    # ---- forward ------------------------------------------------------------------------
    def _alloc_pre(self, cam):
        dev, N, f32, i32 = self.ctx.device, self.N, torch.float32, torch.int32
        return dict(cam=cam, centers=torch.empty((N, 2), dtype=f32, device=dev),
                    depths=torch.empty((N,), dtype=i32, device=dev), colors=torch.empty((N, 3), dtype=f32, device=dev),
                    conics=torch.empty((N, 3), dtype=f32, device=dev),
                    clamped=torch.empty((N, 3), dtype=torch.uint8, device=dev),
                    rect=torch.empty((N, 4), dtype=i32, device=dev), radii=torch.empty((N,), dtype=i32, device=dev))

    def preprocess_batch(self, views, ctx=None, records=None):
        """K1 for several views in one pass over the parameters; returns one record per view to hand to
        forward(view, pre=...).  records=True (default): gs_preprocess_batch_records
        writes each view's 64-byte blend records instead of centers / colors / conics, so no pack pass
        runs per view (the per-view arrays are then None)."""
        c = ctx or self.ctx
        c.sync_stream()
        records = True if records is None else bool(records)   # the blend kernels read records on either list kind
        opac = torch.sigmoid(self.opacity) if self.opacity_is_logit else self.opacity   # once for the batch
        cams = [api.make_camera(v.camera, v.camera_r, v.camera_t, self.result_size, self.tile_size, self.degrees,
                                 self.cam_flags)
                for v in views]
        if records:
            dev, N, i32 = self.ctx.device, self.N, torch.int32
            pres = [dict(cam=cam, centers=None, colors=None, conics=None,
                         rec=torch.empty((N, 16), dtype=torch.float32, device=dev),
                         depths=torch.empty((N,), dtype=i32, device=dev),
                         clamped=torch.empty((N, 3), dtype=torch.uint8, device=dev),
                         rect=torch.empty((N, 4), dtype=i32, device=dev), radii=torch.empty((N,), dtype=i32, device=dev),
                         opac=opac) for cam in cams]
            arr = (GsViewRecords * max(len(pres), 1))()
            for k, p in enumerate(pres):
                arr[k].cam = ctypes.pointer(p["cam"])
                for name, key in (("records", "rec"), ("depths", "depths"), ("clamped", "clamped"),
                                  ("tiles_touched", "rect"), ("radii", "radii")):
                    setattr(arr[k], name, p[key].data_ptr())
            check(c.lib.gs_preprocess_batch_records(c.handle, self.N, api._ptr(self.means), api._ptr(self.log_scales),
                                                    api._ptr(self.quats), api._ptr(self.sh), api._ptr(opac), len(pres),
                                                    arr), "gs_preprocess_batch_records")
            return pres
        pres = [self._alloc_pre(cam) for cam in cams]
        arr = (GsViewPreprocess * max(len(pres), 1))()
        for k, p in enumerate(pres):
            arr[k].cam = ctypes.pointer(p["cam"])
            for name, key in (("centers", "centers"), ("depths", "depths"), ("colors", "colors"), ("conics", "conics"),
                              ("clamped", "clamped"), ("tiles_touched", "rect"), ("radii", "radii")):
                setattr(arr[k], name, p[key].data_ptr())
        check(c.lib.gs_preprocess_batch(c.handle, self.N, api._ptr(self.means), api._ptr(self.log_scales),
                                        api._ptr(self.quats), api._ptr(self.sh), len(pres), arr), "gs_preprocess_batch")
        for p in pres:
            p["opac"] = opac
        return pres

    def forward(self, view, want_keys=False, ctx=None, pre=None, loss=None):
        return self.forward_finish(self.forward_begin(view, ctx=ctx, pre=pre), want_keys=want_keys, loss=loss)

    def forward_begin(self, view, ctx=None, pre=None):
        """K1 (unless `pre` comes from preprocess_batch) and the launch of the tile count; the host
        does not wait.  forward_finish(token) does the rest."""
        c = ctx or self.ctx
        c.sync_stream()
        N = self.N
        if pre is None:
            pre = self._alloc_pre(api.make_camera(view.camera, view.camera_r, view.camera_t, self.result_size,
                                                  self.tile_size, self.degrees, self.cam_flags))
            null = ctypes.c_void_p(0)
            check(c.lib.gs_preprocess(c.handle, N, api._ptr(self.means), api._ptr(self.log_scales),
                                      api._ptr(self.quats), api._ptr(self.sh), ctypes.byref(pre["cam"]),
                                      api._ptr(pre["centers"]), api._ptr(pre["depths"]), api._ptr(pre["colors"]),
                                      api._ptr(pre["conics"]), api._ptr(pre["clamped"]), api._ptr(pre["rect"]),
                                      api._ptr(pre["radii"]), null, null, null, null, null, null), "gs_preprocess")
        opac = pre.get("opac")
        if opac is None:
            opac = torch.sigmoid(self.opacity) if self.opacity_is_logit else self.opacity
        tok = dict(c=c, pre=pre, opac=opac, bin=None)
        if self.cull_tiles:
            # the ctx reads this view's records (or packs the four arrays) while the count is enqueued
            with _records(c, N, pre.get("rec")):
                tok["bin"] = api.bin_culled_begin(c, pre["rect"], pre["depths"], self.result_size, self.tile_size,
                                                  (pre["centers"], pre["colors"], opac, pre["conics"],
                                                   self.pixel_offset))
        else:
            # the reference's own lists (bit-exact keys and ranges): same two halves, no box test
            tok["bin"] = api.bin_exact_begin(c, pre["rect"], pre["depths"], self.result_size, self.tile_size)
        return tok

    def forward_finish(self, tok, want_keys=False, after_blend=None, loss=None):
        """Waits for the tile count, then sort -> K6.  after_blend() (if given) runs right after K6 is
        enqueued: the lane scheduler uses it to launch the lane's NEXT count before this view's loss and K7."""
        self.forward_bin(tok, want_keys=want_keys)
        out = self.forward_blend(tok, loss=loss)
        if after_blend is not None:
            after_blend()
        return out

    def forward_bin(self, tok, want_keys=False):
        """Second stage of a view: waits for its tile count (one host wait), then the sorts and the tile
        ranges, on the ctx the count was launched on."""
        c, pre = tok["c"], tok["pre"]
        c.sync_stream()
        # the records are a per-call binding: whatever raises in between (D >= 2^31, out of memory), the ctx
        # does not keep a pointer to tensors this call owns
        with _records(c, self.N, pre.get("rec")):
            tok["binning"] = api.bin_culled_finish(tok["bin"], want_keys=want_keys)
        return tok["binning"]

    def forward_blend(self, tok, ctx=None, loss=None):
        """Third stage: K6 over the lists of forward_bin, on `ctx` (default: the ctx that binned the view; a
        scheduler that bins on one stream and blends on another passes the blend lane's ctx).
        loss=(target[W,H,3], kind, scale, loss_out[1]): K6 also forms the loss and dL/dimage in its epilogue
        (state["dL"]); loss_out is ADDED to (zero it first)."""
        pre, opac, binning = tok["pre"], tok["opac"], tok["binning"]
        c = ctx or tok["c"]
        c.sync_stream()
        state = dict(ctx=c, cam=pre["cam"], depths=pre["depths"], centers=pre["centers"], colors=pre["colors"],
                     conics=pre["conics"], clamped=pre["clamped"], radii=pre["radii"], opac=opac, binning=binning,
                     rec=pre.get("rec"))
        with _records(c, self.N, pre.get("rec")):
            if loss is None:
                image, n_contrib, t_final = api.blend_forward(c, pre["centers"], pre["colors"], opac, pre["conics"],
                                                              binning, self.bg, self.pixel_offset)
            else:
                # K6 with the loss gradient in its epilogue (gs_blend_forward_loss): no separate pass over the image
                target, kind, scale, loss_out = loss
                image, n_contrib, t_final, dL = api.blend_forward_loss(c, pre["centers"], pre["colors"], opac,
                                                                       pre["conics"], binning, self.bg,
                                                                       self.pixel_offset, target, kind, scale, loss_out)
                state["dL"] = dL
        state["n_contrib"], state["t_final"] = n_contrib, t_final
        self.stats = dict(D=binning.D)
        return image, state

    # ---- loss ---------------------------------------------------------------------------
    def image_loss(self, image, target, kind="l1", scale=1.0, loss_out=None, accumulate=False, ctx=None):
        """Fused loss + dL/dimage (gs_image_loss): kind "l1" = mean|image-target| (the reference's
        metric, train.py:49, computed in float), "l2" = mean squared error.  Returns
        (dL_dimage[W,H,3], loss) with loss a 1-element device tensor (loss_out if given)."""
        c = ctx or self.ctx
        c.sync_stream()
        image = image.contiguous()
        target = target.contiguous()
        if image.shape != target.shape or image.dtype != torch.float32 or target.dtype != torch.float32:
            raise ValueError("image and target must be float32 tensors of the same shape")
        dL = torch.empty_like(image)
        if loss_out is None:
            loss_out = torch.zeros(1, dtype=torch.float32, device=c.device)
        check(c.lib.gs_image_loss(c.handle, image.numel(), api._ptr(image), api._ptr(target),
                                  {"l1": 0, "l2": 1}[kind], float(scale), int(bool(accumulate)), api._ptr(dL),
                                  api._ptr(loss_out)), "gs_image_loss")
        return dL, loss_out

    # ---- backward -----------------------------------------------------------------------
    def backward_rows(self, state, dL_dimage):
        """K7 for one view: leaves the blend gradients in state["rows"] ([N,16], the kernel's
        64-byte rows) for backward_params.  Runs on the ctx (lane) the view was rendered on."""
        c = state.get("ctx", self.ctx)
        c.sync_stream()
        b = state["binning"]
        W, H = self.result_size
        rows = torch.empty((self.N, 16), dtype=torch.float32, device=c.device)
        with _records(c, self.N, state.get("rec")):
            check(c.lib.gs_blend_backward_rows(
                c.handle, self.N, api._ptr(state["centers"]), api._ptr(state["colors"]), api._ptr(state["opac"]),
                api._ptr(state["conics"]), api._ptr(b.vals), api._ptr(b.ranges), b.D, ctypes.byref(self.bg), W, H,
                self.tile_size, self.pixel_offset, api._ptr(state["n_contrib"]), api._ptr(state["t_final"]),
                api._ptr(dL_dimage.contiguous()), api._ptr(rows)), "gs_blend_backward_rows")
        state["rows"] = rows
        return rows

    def backward_params(self, states, grads, accumulate=False, ctx=None, g_range=None):
        """K8 over every view in `states` at once (each must have been through backward_rows):
        parameters are read once and the gradients written (or added) once.  g_range=(g0, g1) restricts
        the launch to those gaussians (every array of the call is per-gaussian and contiguous, so a range
        is the same call on offset pointers); g0 must be a multiple of 4 (16-byte aligned rows of every
        array)."""
        c = ctx or self.ctx
        c.sync_stream()
        g0, g1 = (0, self.N) if g_range is None else (int(g_range[0]), int(g_range[1]))
        if not (0 <= g0 <= g1 <= self.N) or g0 % 4:
            raise ValueError(f"g_range {g_range} must lie in [0, {self.N}] and start at a multiple of 4")
        at = lambda t, width: ctypes.c_void_p(t.data_ptr() + g0 * width * t.element_size())
        views = (GsViewBackward * max(len(states), 1))()
        for k, st in enumerate(states):
            views[k].cam = ctypes.pointer(st["cam"])
            views[k].radii = at(st["radii"], 1).value
            views[k].clamped = at(st["clamped"], 3).value
            views[k].grad_rows = at(st["rows"], 16).value
        check(c.lib.gs_backward_params_batch(
            c.handle, g1 - g0, at(self.means, 3), at(self.log_scales, 3), at(self.quats, 4),
            at(self.sh, 48), at(self.opacity, 1), int(self.opacity_is_logit), len(states), views,
            int(bool(accumulate)), at(grads.means, 3), at(grads.log_scales, 3), at(grads.quats, 4),
            at(grads.opacity, 1), at(grads.sh, 48)), "gs_backward_params_batch")
        return grads

    def backward(self, state, dL_dimage, grads, accumulate=True):
        """d(loss)/d(parameters) of one view into `grads` (FlatGrads), added when accumulate."""
        self.backward_rows(state, dL_dimage)
        return self.backward_params([state], grads, accumulate=accumulate)


def shard_views(n_views, rank, world_size):
    """Round-robin view -> rank assignment (view v goes to rank v % world_size)."""
    return [v for v in range(n_views) if v % world_size == rank]


# main.py:11-17, in the order of the flat gradient buffer (means, log_scales, quats, opacity, sh)
REFERENCE_LEARNING_RATES = {"center": .0008, "scaling": .025, "rotation": .005, "opacity": .25, "sh": .0125}


class Optimizer:
    """Parameter update on the flat [N*59] gradient buffer (gs_adam_step).  method "adam": Adam with
    one learning rate per parameter group; "reference": the intent of train.py:51-56,
    param -= lr * grad * loss."""

    def __init__(self, pipe, learning_rates=None, method="adam", betas=(0.9, 0.999), eps=1e-15):
        lr = dict(REFERENCE_LEARNING_RATES)
        lr.update(learning_rates or {})
        self.pipe, self.method, self.betas, self.eps = pipe, method, betas, float(eps)
        self.lr = (ctypes.c_float * 5)(lr["center"], lr["scaling"], lr["rotation"], lr["opacity"], lr["sh"])
        self.step_count = 0
        n = flat_offsets(pipe.N)[1]
        dev = pipe.ctx.device
        self.m = torch.zeros(n, dtype=torch.float32, device=dev) if method == "adam" else None
        self.v = torch.zeros(n, dtype=torch.float32, device=dev) if method == "adam" else None

    def step(self, grads, loss=1.0):
        p, c = self.pipe, self.pipe.ctx
        c.sync_stream()
        self.step_count += 1
        check(c.lib.gs_adam_step(c.handle, p.N, api._ptr(p.means), api._ptr(p.log_scales), api._ptr(p.quats),
                                 api._ptr(p.opacity), api._ptr(p.sh), api._ptr(grads.flat), api._ptr(self.m),
                                 api._ptr(self.v), ctypes.byref(self.lr), float(self.betas[0]), float(self.betas[1]),
                                 self.eps, self.step_count, 0 if self.method == "adam" else 1, float(loss)),
              "gs_adam_step")


def _loss_and_rows(pipe, image, state, target, loss_kind, scale, loss_out):
    """Loss gradient (unless K6 formed it: state["dL"]) and K7 of one rendered view, on the current stream."""
    dL = state.pop("dL", None)
    if dL is None:
        dL, _ = pipe.image_loss(image, target, loss_kind, scale, loss_out, ctx=state["ctx"])
    pipe.backward_rows(state, dL)


def _render_on_lanes(pipe, views, mine, targets, losses, n_lanes, loss_kind, scale, fused_loss=False):
    """Forward, loss gradient and K7 of this rank's views on n_lanes concurrent lanes.  Returns
    (states, done events per view, lanes)."""
    main = pipe.ctx.current_stream()
    # a process has 4 hardware queues (GPU_MAX_HW_QUEUES): up to 3 side lanes, the main stream is the 4th
    if n_lanes > 3:
        lanes = pipe.lanes(n_lanes - 1) + [(pipe.ctx, main)]
    else:
        lanes = pipe.lanes(n_lanes)
    # K1 for the first view of every lane, then (on the main stream, under those views' binning) for the rest
    first = min(n_lanes, len(mine))
    pres = pipe.preprocess_batch([views[v] for v in mine[:first]])
    for _, stream in lanes:
        stream.wait_stream(main)
    pre_ready = None
    if len(mine) > first:
        pres += pipe.preprocess_batch([views[v] for v in mine[first:]])
        pre_ready = main.record_event()
    # Two workspaces per lane: the tile count of a lane's next view is launched right behind this
    # view's K6, so its result is on the host long before the lane has finished K7 and the host
    # never waits for a whole view (a lane's queue does not run dry between views).
    alts = pipe.alt_contexts(n_lanes)
    tokens = {}

    def begin(k):
        if k >= len(mine):
            return
        ctx, stream = lanes[k % n_lanes]
        if (k // n_lanes) & 1:
            ctx = alts[k % n_lanes]
        with pipe.ctx.use_stream(stream):
            if k >= first:
                stream.wait_event(pre_ready)
            tokens[k] = pipe.forward_begin(views[mine[k]], ctx=ctx, pre=pres[k])

    for k in range(first):
        begin(k)
    states, done = [], []
    # host order per round of n_lanes views: sort + K6 (+ the lane's next count) for every lane, then
    # loss + K7 for every lane
    for r0 in range(0, len(mine), n_lanes):
        ks = range(r0, min(r0 + n_lanes, len(mine)))
        fwd = {}
        for k in ks:
            with pipe.ctx.use_stream(lanes[k % n_lanes][1]):
                la = (targets[mine[k]], loss_kind, scale, losses[k:k + 1]) if fused_loss else None
                fwd[k] = pipe.forward_finish(tokens.pop(k), after_blend=lambda k=k: begin(k + n_lanes), loss=la)
        for k in ks:
            stream = lanes[k % n_lanes][1]
            image, state = fwd[k]
            with pipe.ctx.use_stream(stream):
                _loss_and_rows(pipe, image, state, targets[mine[k]], loss_kind, scale, losses[k:k + 1])
                done.append(stream.record_event())
            states.append(state)
    return states, done, lanes


def _render_prepared(pipe, views, mine, targets, losses, n_lanes, loss_kind, scale, fused_loss=False):
    """The same work as _render_on_lanes with the front half of every view's lists -- count, depth order, scanned
    offsets -- done per BATCH of views on the main stream (gs_bin_prepare_begin / _end: a dozen launches and one host
    wait per batch instead of per view) and only what scales with the number of pairs -- emit, tile sort, ranges
    (gs_bin_fill_prepared) -- on the view's lane, in front of its blend kernels.  The first batch is the first view of
    every lane, so the lanes start after one short K1; the second batch (the rest, up to 8 views) is projected and
    counted under the first views' blends.  Returns (states, done events per view, lanes)."""
    main = pipe.ctx.current_stream()
    if n_lanes > 3:
        lanes = pipe.lanes(n_lanes - 1) + [(pipe.ctx, main)]
    else:
        lanes = pipe.lanes(n_lanes)
    prep = pipe.alt_contexts(2)          # workspaces of the batches in flight; both launch on the main stream
    first = min(n_lanes, len(mine))
    groups = [list(range(first))] + [list(range(g0, min(g0 + 8, len(mine)))) for g0 in range(first, len(mine), 8)]
    pres, toks, ready = {}, {}, {}
    group_of = {k: gi for gi, g in enumerate(groups) for k in g}

    def start(gi):
        ks = groups[gi]
        for k, p in zip(ks, pipe.preprocess_batch([views[mine[k]] for k in ks])):
            pres[k] = p
        toks[gi] = api.bin_prepare_begin(prep[gi % 2], [pres[k] for k in ks], pipe.result_size, pipe.tile_size,
                                         pipe.pixel_offset, pipe.cull_tiles)

    def finish(gi):
        api.bin_prepare_end(toks.pop(gi))           # the one host wait of the batch
        ready[gi] = main.record_event()

    start(0)
    finish(0)
    if len(groups) > 1:
        start(1)                                    # K1 + counts of the later views, behind the first batch's sorts

    def advance():
        """Finishes the batch whose counts are in flight and starts the one after it.  Its sorts and scans go onto the
        main stream, so this runs BEFORE the main stream receives a view's blend kernels (about a millisecond of
        work that the other lanes' next views would otherwise wait behind)."""
        for gi in sorted(toks):
            finish(gi)
            if gi + 1 < len(groups) and gi + 1 not in ready:
                start(gi + 1)

    states, done = [], []
    for r0 in range(0, len(mine), n_lanes):
        # side lanes first, the main stream's view last
        ks = sorted(range(r0, min(r0 + n_lanes, len(mine))), key=lambda k: (lanes[k % n_lanes][1] is main, k))
        fwd = {}
        for k in ks:
            ctx, stream = lanes[k % n_lanes]
            if stream is main:
                advance()
            while group_of[k] not in ready:         # more than two batches: the later ones start as the earlier finish
                if not toks:
                    start(min(g for g in range(len(groups)) if g not in ready))
                advance()
            with pipe.ctx.use_stream(stream):
                stream.wait_event(ready[group_of[k]])
                with _records(ctx, pipe.N, pres[k].get("rec")):
                    b = api.bin_fill_prepared(ctx, pres[k])
                la = (targets[mine[k]], loss_kind, scale, losses[k:k + 1]) if fused_loss else None
                fwd[k] = pipe.forward_blend(dict(pre=pres[k], opac=pres[k]["opac"], binning=b, c=ctx), loss=la)
        ks = sorted(ks)
        for k in ks:
            stream = lanes[k % n_lanes][1]
            image, state = fwd[k]
            with pipe.ctx.use_stream(stream):
                _loss_and_rows(pipe, image, state, targets[mine[k]], loss_kind, scale, losses[k:k + 1])
                done.append(stream.record_event())
            states.append(state)
    return states, done, lanes


def _render_batched(pipe, views, mine, targets, losses, n_lanes, loss_kind, scale, group):
    """The same work as _render_on_lanes with the lists of `group` views built per batch of launches on the main
    stream (gs_bin_culled_batch: no per-view host wait, about 20 launches per group instead of 25 per view) and the
    blend kernels of each view on one of n_lanes - 1 side streams, behind its group's event.  One host wait per
    step: the rect-pair bounds that size the lists.  Returns (states, done events per view, lanes)."""
    main = pipe.ctx.current_stream()
    lanes = pipe.lanes(n_lanes - 1) if n_lanes > 1 else [(pipe.ctx, main)]
    pres = pipe.preprocess_batch([views[v] for v in mine], records=True)
    bounds = []
    for g0 in range(0, len(pres), 8):
        bounds += api.bin_bounds_batch(pipe.ctx, pres[g0:g0 + 8], pipe.result_size, pipe.tile_size)
    bits = max(max([b[1] for b in bounds] + [1]).bit_length(), 1)
    states, done = [None] * len(mine), [None] * len(mine)
    for g0 in range(0, len(mine), group):
        ks = list(range(g0, min(g0 + group, len(mine))))
        cap = max([bounds[k][0] for k in ks] + [1])
        bins = api.bin_culled_batch(pipe.ctx, [pres[k] for k in ks], pipe.result_size, pipe.tile_size, pipe.pixel_offset,
                                    depth_bits=bits, capacity=cap)
        ready = main.record_event()
        for k, b in zip(ks, bins):
            ctx, stream = lanes[k % len(lanes)]
            with pipe.ctx.use_stream(stream):
                stream.wait_event(ready)
                image, state = pipe.forward_blend(dict(pre=pres[k], opac=pres[k]["opac"], binning=b, c=ctx))
                dL, _ = pipe.image_loss(image, targets[mine[k]], loss_kind, scale, losses[k:k + 1], ctx=ctx)
                pipe.backward_rows(state, dL)
                done[k] = stream.record_event()
            states[k] = state
    return states, done, lanes


def gaussian_chunks(N, n_chunks):
    """[g0, g1) ranges covering N gaussians in about n_chunks pieces; every g0 is a multiple of 256."""
    step = max(256, (-(-N // max(int(n_chunks), 1)) + 255) // 256 * 256)
    return [(g0, min(g0 + step, N)) for g0 in range(0, N, step)] or [(0, 0)]


def all_reduce_sum(ctx, tensors, works=None):
    """SUM all-reduce of every tensor in `tensors` over the default process group -- the ONE place the package
    issues a collective.  RCCL ("nccl"): stream-ordered behind what the current stream holds and run on the
    communicator's own stream; with a `works` list they are issued asynchronously and the handles appended (wait on
    them before the result is used), otherwise the call blocks the stream as usual.  Host-staged back ends (gloo):
    always behind a device synchronisation and blocking -- measured on MI355X with two ranks sharing one GPU, a
    gloo all-reduce of a device tensor issued while the rank's lane streams still held work never completed (the
    same collective alone, or behind a synchronisation, completes in 135 ms for 236 MB)."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        for t in tensors:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=works is not None)
            if works is not None:
                works.append(w)
        return
    if any(t.device.type != "cpu" for t in tensors):
        ctx.synchronize()
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _all_reduce(pipe, tensors, works):
    all_reduce_sum(pipe.ctx, tensors, works)


def grad_all_reduce_alone(pipe, grads, allreduce_chunks=4):
    """The collectives of one train_step without its kernels (the same slices in the same order): what bench.py times
    on an idle GPU to say how much of a multi-rank step is the gradient exchange."""
    offs, _ = flat_offsets(pipe.N)
    works = []
    for g0, g1 in gaussian_chunks(pipe.N, allreduce_chunks):
        _all_reduce(pipe, [grads.flat[offs[4] + 48 * g0:offs[4] + 48 * g1]], works)
    _all_reduce(pipe, [grads.flat[:offs[4]]], works)
    for w in works:
        w.wait()


BIG_VIEW_PAIRS = 32 << 20     # (gaussian, tile) pairs per view from which two lanes beat four (measured, 10 M / 4K)


def auto_lanes(pipe):
    """Views in flight for train_step(concurrent_views=None): four lanes while a view's lists are small (the blend
    kernels' tails overlap), two once a view carries tens of millions of pairs -- its sorts are then HBM-bound for
    milliseconds and four of them at once only stretch each other (10 M gaussians at 3840x2160, 122 M pairs per
    view, 4 views per step: 22.99 / 21.77 / 22.16 / 22.90 ms with 1 / 2 / 3 / 4 lanes)."""
    return 2 if int(pipe.stats.get("D", 0)) >= BIG_VIEW_PAIRS else 4


def train_step(pipe, views, targets, grads, loss_scale=None, concurrent_views=2, loss_kind="l2", states_hook=None,
               k8_early_lanes=None, allreduce_chunks=4, batch_binning=False, bin_group=4, prepared=False,
               fused_loss=False):
    """Forward+backward over this rank's shard of `views`, then the all-reduce of the flat gradient
    buffer when torch.distributed is initialised.  targets[v] is the image [W,H,3] view v should
    match; the loss is the mean squared ("l2") or mean absolute ("l1") error summed over the views,
    formed together with dL/dimage by the fused native kernel.  concurrent_views > 1 renders that
    many local views at a time on separate streams (the 4th is the main stream; None: auto_lanes, by the size of
    the last view's lists).  The views of the
    first k8_early_lanes lanes (default: half of them) get their parameter backward as soon as they are
    complete, under the blend kernels of the lanes still running.  With more than one rank the last
    parameter-backward launch runs over `allreduce_chunks` ranges of gaussians, and the SH gradients of a
    range (48 of the 59 floats per gaussian) are all-reduced while the next range is computed; the other
    four groups follow as one slice.  batch_binning=True builds the lists of bin_group views per batch of
    launches with no per-view host wait (gs_bin_culled_batch; one wait per step for the bounds) -- measured on
    MI355X it does not pay at 1080p / 1M (6.18 ms per 8-view step against 5.92 ms: the batched sorts are 25 %
    cheaper alone, but the step loses the overlap of one view's sorts with another's blends), so the default is
    the per-view path.  prepared=True: the count, depth order and scanned offsets of the views are built per batch of
    views on the main stream (gs_bin_prepare_begin / _end) and only emit + tile sort + ranges run on the view's lane
    (_render_prepared); False (default): every view builds its lists on its lane (_render_on_lanes).  Measured on
    MI355X at 1 M / 1080p / 8 views: 6.02-6.13 ms per step prepared (and 7.9 ms in some launch arrangements: the
    main stream then carries the batches' sorts AND a lane's blends) against 5.85-5.89 ms per view, so it is off.
    fused_loss=True: K6 forms the loss and dL/dimage in its epilogue (gs_blend_forward_loss): one pass over the image
    and two launches per view less; False (default): the separate gs_image_loss pass.  Measured on MI355X at 1 M /
    1080p / 8 views: 5.91 ms fused against 5.82 ms -- the epilogue's 12-byte-per-lane target reads and dL stores are
    worse streams than the loss kernel's float4s, and that outweighs the saved pass.
    Returns (sum of local losses as a device scalar, number of local views)."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    mine = shard_views(len(views), rank, world)
    scale = 1.0 if loss_scale is None else loss_scale
    states = []
    if concurrent_views is None:
        concurrent_views = auto_lanes(pipe)
    n_lanes = min(int(concurrent_views), len(mine))
    losses = torch.zeros(max(len(mine), 1), dtype=torch.float32, device=pipe.ctx.device)
    late = None
    # the ordered reduction of K7 works on a list of known length: per-view lists (one host wait each)
    if getattr(pipe.ctx, "options", {}).get("deterministic_backward"):
        batch_binning = False
    if batch_binning and pipe.cull_tiles and len(mine) > 0:
        main = pipe.ctx.current_stream()
        group = max(1, min(int(bin_group), 8))
        states, done, lanes = _render_batched(pipe, views, mine, targets, losses, n_lanes, loss_kind, scale, group)
        # the views of all groups but the last get their parameter backward on the main stream (idle once the
        # lists are built) while the last group still blends
        n_early = (len(mine) - 1) // group * group
        if states_hook is None and n_lanes > 1 and 2 <= n_early and k8_early_lanes != 0:
            for k in range(n_early):
                main.wait_event(done[k])
            pipe.backward_params(states[:n_early], grads, accumulate=False)
            late = states[n_early:]
        for _, stream in lanes:
            if stream is not main:
                main.wait_stream(stream)
    elif n_lanes > 1:
        main = pipe.ctx.current_stream()
        render = _render_prepared if prepared else _render_on_lanes
        states, done, lanes = render(pipe, views, mine, targets, losses, n_lanes, loss_kind, scale, fused_loss)
        # The lanes do not finish together (the hardware serves the queues in a fixed order): K8 of the
        # early lanes' views runs on lane 0's stream while the others still blend.
        if k8_early_lanes is None:
            k8_early_lanes = n_lanes // 2 if n_lanes > 3 else 0
        early = [k for k in range(len(mine)) if k % n_lanes < k8_early_lanes]
        if states_hook is None and 2 <= len(early) < len(mine):
            ctx0, stream0 = lanes[0]
            with pipe.ctx.use_stream(stream0):
                for k in early:
                    stream0.wait_event(done[k])
                pipe.backward_params([states[k] for k in early], grads, accumulate=False, ctx=ctx0)
            late = [st for k, st in enumerate(states) if k not in set(early)]
        for _, stream in lanes:
            main.wait_stream(stream)
    else:
        pres = pipe.preprocess_batch([views[v] for v in mine])
        for k, v in enumerate(mine):
            la = (targets[v], loss_kind, scale, losses[k:k + 1]) if fused_loss else None
            image, state = pipe.forward(views[v], pre=pres[k], loss=la)
            _loss_and_rows(pipe, image, state, targets[v], loss_kind, scale, losses[k:k + 1])
            states.append(state)
    if states_hook is not None:      # e.g. Densifier.accumulate: per-view blend gradients are still alive here
        states_hook(states)
    # one pass over the gaussians for the (remaining) local views; writes, or adds to the early launch
    last, accumulate = (states, False) if late is None else (late, True)
    if not distributed:          # (a one-rank group takes the collective path too: it costs three more launches)
        pipe.backward_params(last, grads, accumulate=accumulate)
    else:
        offs, _ = flat_offsets(pipe.N)
        works = []
        for g0, g1 in gaussian_chunks(pipe.N, allreduce_chunks):
            pipe.backward_params(last, grads, accumulate=accumulate, g_range=(g0, g1))
            _all_reduce(pipe, [grads.flat[offs[4] + 48 * g0:offs[4] + 48 * g1]], works)
        _all_reduce(pipe, [grads.flat[:offs[4]]], works)      # means, log-scales, quaternions, opacity: 11 floats each
        for w in works:
            w.wait()                                          # the current stream waits for the communicator's
    loss = losses.sum()
    return loss, len(mine)
