"""Single-stream duration of every kernel of one forward+backward view of the bench scene, measured with the
library's own HIP events (gs_ctx_profile): nothing else runs on the GPU, so these are the figures a
rocprofv3 --kernel-trace of the same script shows.  GS_VIEWS views (default 8), GS_N gaussians.
    python tools/kernel_times.py [--exact]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsplat_amd as gs
from bench import make_scene, make_views


def measure(pipe, views, target, n_views, repeats=3):
    ctx = pipe.ctx
    grads = gs.FlatGrads(pipe.N, ctx.device)
    out = {}
    for rep in range(repeats + 1):
        if rep == 1:
            ctx.profile(True)
            ctx.read_profile()
        states = []
        pres = pipe.preprocess_batch(views[:n_views])
        for v in range(n_views):
            image, st = pipe.forward(views[v], pre=pres[v])
            dL, _ = pipe.image_loss(image, target, "l2", 1.0)
            pipe.backward_rows(st, dL)
            states.append(st)
        pipe.backward_params(states, grads)
        torch.cuda.synchronize()
    for k, (ms, cnt) in ctx.read_profile().items():
        if cnt:
            out[k] = {"ms_per_launch": ms / cnt, "launches": int(cnt)}
    ctx.profile(False)
    return out, states[-1]


if __name__ == "__main__":
    N = int(os.environ.get("GS_N", 1_000_000))
    W, H = int(os.environ.get("GS_W", 1920)), int(os.environ.get("GS_H", 1080))
    ls = (float(os.environ.get("GS_LS0", 1.0)), float(os.environ.get("GS_LS1", 3.0)))
    g, cam = make_scene(N, W, H, ls)
    ctx, _, _ = gs.setup_gpu(0)
    pipe = gs.SplatPipeline(g, [W, H], ctx=ctx, cull_tiles="--exact" not in sys.argv)
    n = int(os.environ.get("GS_VIEWS", 8))
    views = make_views(gs, cam, 8)
    target = torch.rand(W, H, 3, device=ctx.device)
    times, st = measure(pipe, views, target, n)
    print(json.dumps({k: round(v["ms_per_launch"], 4) for k, v in sorted(times.items())}))
    print("sum per view (ms):", round(sum(v["ms_per_launch"] * v["launches"] for v in times.values()) / (3 * n), 4))
