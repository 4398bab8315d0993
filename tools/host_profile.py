"""Where the host time of one training step goes (cProfile over a few steps of the bench configuration)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsplat_amd as gs
from bench import make_scene, make_views
N, W, H = 1_000_000, 1920, 1080
g, cam = make_scene(N, W, H, (1.0, 3.0))
ctx, _, _ = gs.setup_gpu(0)
pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
views = make_views(gs, cam, 8)
grads = gs.FlatGrads(N, ctx.device)
targets = [torch.rand(W, H, 3, device=ctx.device)] * 8
for _ in range(5):
    gs.train_step(pipe, views, targets, grads, concurrent_views=4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    gs.train_step(pipe, views, targets, grads, concurrent_views=4)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
