"""Times the tile sort alone on the 10 M / 4K shape for different pass counts (experiment)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("gaussian-splatting_amd._lib")
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libgsplat_exp.so")
import torch
import gsplat_amd as gs
from bench import make_scene, make_views
N, W, H = int(os.environ.get("GS_N", 4_000_000)), 3840, 2160
g, cam = make_scene(N, W, H, (2.0, 4.0))
ctx, _, _ = gs.setup_gpu(0)
pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
views = make_views(gs, cam, 8)
for rep in range(3):
    if rep == 1:
        ctx.profile(True); ctx.read_profile()
    image, st = pipe.forward(views[rep % 2])
torch.cuda.synchronize()
prof = ctx.read_profile()
print(os.environ.get("GS_EXP_TILE_PASSES"), "D", st["binning"].D, {k: round(v[0] / v[1], 3) for k, v in prof.items() if v[1] and k.startswith("bin")})
