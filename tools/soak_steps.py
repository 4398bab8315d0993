"""Soak test of the four-lane training step: GS_STEPS steps of the bench scene, every step's gradient
compared with the first step's.  The atomics make the float summation order vary, so the difference
is not zero, but a race between lanes (a workspace reused too early, a count read before it is
written) would show as a large, sporadic one.  Prints the largest relative difference seen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsplat_amd as gs
from bench import make_scene, make_views

N = int(os.environ.get("GS_N", 1_000_000))
STEPS = int(os.environ.get("GS_STEPS", 300))
CHECK_EVERY = int(os.environ.get("GS_CHECK_EVERY", 1))   # >1: steps run back to back, as in bench.py, between checks
W, H = 1920, 1080
g, cam = make_scene(N, W, H, (1.0, 3.0))
ctx, _, _ = gs.setup_gpu(0)
pipe = gs.SplatPipeline(g, [W, H], opacity_is_logit=True, ctx=ctx)
views = make_views(gs, cam, 8)
grads = gs.FlatGrads(N, ctx.device)
gen = torch.Generator(device="cpu").manual_seed(1234)
target = torch.rand(W, H, 3, generator=gen).to(ctx.device)
targets = [target] * len(views)

first, worst, worst_step, losses = None, 0.0, -1, set()
for s in range(STEPS):
    loss, _ = gs.train_step(pipe, views, targets, grads, concurrent_views=4)
    if s % CHECK_EVERY:
        continue
    torch.cuda.synchronize()
    losses.add(round(float(loss), 3))
    flat = grads.flat
    assert torch.isfinite(flat).all(), f"step {s}: non-finite gradient"
    if first is None:
        first = flat.clone()
        scale = first.abs().max().item()
        continue
    d = (flat - first).abs().max().item() / scale
    if d > worst:
        worst, worst_step = d, s
    if s % 50 == 0:
        print(f"step {s}: worst so far {worst:.3e} (step {worst_step})", flush=True)
print(f"{STEPS} steps: max |grad - grad_0| / max|grad_0| = {worst:.3e} at step {worst_step}; "
      f"distinct losses (3 decimals): {sorted(losses)}")
assert worst < 1e-4, "steps disagree beyond float summation order"
