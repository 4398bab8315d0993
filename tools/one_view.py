"""A few forward+backward views of the bench scene on ONE stream (for rocprofv3 --pmc passes, where
per-kernel counters must not overlap).  GS_VIEWS views, then one batched parameter backward.  GS_LISTS = exact
(default, bench.py's default) or culled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
if os.environ.get("GS_VARIANT"):      # a tools/libgsplat_<name>.so built by variant_bench.py
    _lib = importlib.import_module("gaussian-splatting_amd._lib")
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgsplat_%s.so" % os.environ["GS_VARIANT"])
import torch
import gsplat_amd as gs
from bench import make_scene, make_views

N = int(os.environ.get("GS_N", 1_000_000))
W, H = int(os.environ.get("GS_W", 1920)), int(os.environ.get("GS_H", 1080))
g, cam = make_scene(N, W, H, (float(os.environ.get("GS_LS0", 1.0)), float(os.environ.get("GS_LS1", 3.0))))
ctx, _, _ = gs.setup_gpu(0)
LISTS = os.environ.get("GS_LISTS", "exact")     # bench.py's default: the reference's own lists
pipe = gs.SplatPipeline(g, [W, H], ctx=ctx, cull_tiles=LISTS == "culled")
views = make_views(gs, cam, 8)
grads = gs.FlatGrads(N, ctx.device)
target = torch.rand(W, H, 3, device=ctx.device)
states = []
n = int(os.environ.get("GS_VIEWS", 2))
pres = pipe.preprocess_batch(views[:n])
for v in range(n):
    image, st = pipe.forward(views[v], pre=pres[v])
    dL, _ = pipe.image_loss(image, target, "l2", 1.0)
    pipe.backward_rows(st, dL)
    states.append(st)
pipe.backward_params(states, grads)
torch.cuda.synchronize()
print("done", st["binning"].D)
