"""BASELINE.json configs[3]: a Mip-NeRF360-scale scene with a GROWING gaussian count -- training steps at ~1300x840
starting from 3 M gaussians, densification (clone / split / prune) until >= 5 M, with the workspace of every context
reserved ahead (SplatPipeline.reserve), so that no step regrows a buffer.  Prints one JSON line: ms per step and
workspace bytes before / after each densification, and the library's allocation count (constant after the reserve).
    python tools/bench_config3.py [--gaussians 3000000] [--target 5000000] [--views 4] [--steps 4]
tests/test_gpu_config3.py runs the same routine with fewer steps and asserts on its record."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(gpu_ctx=None, n0=3_000_000, target=5_000_000, W=1300, H=840, n_views=4, steps=3, lanes=None, log_scale=(1.0, 3.0),
        lists="exact"):
    import torch
    import gsplat_amd as gs
    from bench import make_scene, make_views
    ctx = gpu_ctx or gs.setup_gpu(0)[0]
    g, cam = make_scene(n0, W, H, log_scale)
    pipe = gs.SplatPipeline(g, [W, H], opacity_is_logit=True, ctx=ctx, cull_tiles=lists == "culled")
    views = make_views(gs, cam, n_views)
    gen = torch.Generator(device="cpu").manual_seed(7)
    targets = [torch.rand(W, H, 3, generator=gen).to(ctx.device)] * n_views
    opt = gs.Optimizer(pipe, None)
    # densify every gaussian that is "hot" (threshold 0: all visible ones), capped by a budget that grows per round
    den = gs.Densifier(pipe, opt, grad_threshold=0.0, percent_dense=0.01, scene_extent=3.0, seed=3)   # big: scale > 0.03
    n_max = int(target * 1.12)
    _, st = pipe.forward(views[0])
    d0 = st["binning"].D
    d_max = int(d0 * (n_max / n0) * 1.3)
    del st
    pipe.reserve(n_max, d_max, concurrent_views=lanes or 4)
    rec = {"workload": f"configs[3]: {W}x{H}, {n_views} views per step, start {n0} gaussians, densify until >= {target}",
           "lists": lists, "reserved": {"n_max": n_max, "d_max": d_max, "workspace_bytes": pipe.workspace_bytes()},
           "rounds": []}
    alloc0 = pipe.allocations()
    budgets = [int(n0 + (target - n0) * 0.55), int(target * 1.04), int(target * 1.1)]
    for rnd in range(len(budgets) + 1):
        grads = gs.FlatGrads(pipe.N, ctx.device)
        ms = []
        for s in range(steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss, _ = gs.train_step(pipe, views, targets, grads, loss_scale=1.0 / n_views, concurrent_views=lanes,
                                    states_hook=den.accumulate)
            opt.step(grads)
            torch.cuda.synchronize()
            if s:                       # the first step after a change of N is the warm-up
                ms.append((time.perf_counter() - t0) * 1e3)
        r = {"N": pipe.N, "D_view0": int(pipe.stats.get("D", 0)), "ms_per_step": round(sum(ms) / len(ms), 3),
             "loss": float(loss), "workspace_bytes": pipe.workspace_bytes(), "allocations_since_reserve": pipe.allocations() - alloc0,
             "torch_allocated_bytes": int(torch.cuda.memory_allocated(ctx.device))}
        rec["rounds"].append(r)
        if pipe.N >= target or rnd == len(budgets):
            break
        den.max_gaussians = budgets[rnd]
        t0 = time.perf_counter()
        stats = den.step()
        torch.cuda.synchronize()
        r["densify"] = dict(stats, ms=round((time.perf_counter() - t0) * 1e3, 1))
    rec["final_N"] = pipe.N
    rec["finite"] = bool(torch.isfinite(pipe.means).all()) and bool(torch.isfinite(grads.flat).all())
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=3_000_000)
    ap.add_argument("--target", type=int, default=5_000_000)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--lists", default="exact")
    a = ap.parse_args()
    print(json.dumps(run(None, a.gaussians, a.target, n_views=a.views, steps=a.steps, lists=a.lists)), flush=True)
