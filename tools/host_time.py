import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gsplat_amd as gs
from bench import make_scene, make_views
N, W, H = 1_000_000, 1920, 1080
g, cam = make_scene(N, W, H, (1.0, 3.0))
ctx, _, _ = gs.setup_gpu(0)
pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
views = make_views(gs, cam, 8)
grads = gs.FlatGrads(N, ctx.device)
target = torch.rand(W, H, 3, device=ctx.device)
targets = [target] * 8
for _ in range(5):
    gs.train_step(pipe, views, targets, grads, concurrent_views=4)
torch.cuda.synchronize()
for trial in range(3):
    host, tot = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gs.train_step(pipe, views, targets, grads, concurrent_views=4)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print("host issue %.2f ms, step incl. drain %.2f ms" % (1e3 * sum(host) / 10, 1e3 * sum(tot) / 10))
