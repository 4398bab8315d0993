"""Per-stage durations of gs_bin_culled_batch (library events, nothing else on the GPU) for batches of 1, 4 and 8
views of the bench scene, beside the per-view entry points.
    python tools/bin_batch_times.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsplat_amd as gs
from bench import make_scene, make_views


def main():
    N, W, H = int(os.environ.get("GS_N", 1_000_000)), 1920, 1080
    g, cam = make_scene(N, W, H, (1.0, 3.0))
    ctx, _, _ = gs.setup_gpu(0)
    pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
    views = make_views(gs, cam, 8)
    pres = pipe.preprocess_batch(views, records=True)
    bounds = gs.api.bin_bounds_batch(ctx, pres, [W, H], 16)
    cap = max(b[0] for b in bounds)
    bits = max(b[1] for b in bounds).bit_length()
    print("rect pairs", [b[0] for b in bounds], "depth bits", bits)
    for nv in (1, 4, 8):
        for rep in range(3):
            if rep == 1:
                ctx.profile(True); ctx.read_profile()
            out = gs.api.bin_culled_batch(ctx, pres[:nv], [W, H], 16, 0.5, depth_bits=bits, capacity=cap)
        prof = ctx.read_profile()
        ctx.profile(False)
        per = {k: round(v[0] / 2 / nv, 4) for k, v in prof.items() if v[1]}
        print(f"batch of {nv}: ms per view", per, "sum", round(sum(per.values()), 4), "D", out[0].true_D(), flush=True)
    for rep in range(3):
        if rep == 1:
            ctx.profile(True); ctx.read_profile()
        for v in range(4):
            tok = pipe.forward_begin(views[v], pre=pres[v])
            pipe.forward_bin(tok)
    prof = ctx.read_profile()
    per = {k: round(v[0] / 8, 4) for k, v in prof.items() if v[1]}
    print("per-view calls: ms per view", per, "sum", round(sum(per.values()), 4))


if __name__ == "__main__":
    main()
