"""train_step time of the bench workload under scheduler variants: "bG,L" = lists built per batch of G views
(gs_bin_culled_batch) and L streams; "p,L" = per-view lists on L lanes.
    python tools/step_time.py b4,4 b2,4 b8,4 p,4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("GS_LIB"):        # a variant library built by tools/variant_bench.py build
    _lib = __import__("importlib").import_module("gaussian-splatting_amd._lib")
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgsplat_%s.so" % os.environ["GS_LIB"])
import torch
import gsplat_amd as gs
from bench import make_scene, make_views
pl = __import__("importlib").import_module("gaussian-splatting_amd.pipeline")


def main():
    N, W, H = int(os.environ.get("GS_N", 1_000_000)), 1920, 1080
    g, cam = make_scene(N, W, H, (1.0, 3.0))
    ctx, _, _ = gs.setup_gpu(0)
    pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
    nv = int(os.environ.get("GS_VIEWS", 8))
    views = make_views(gs, cam, 8)[:nv]
    targets = [torch.rand(W, H, 3, device=ctx.device)] * nv
    grads = gs.FlatGrads(N, ctx.device)
    ref = None
    for variant in sys.argv[1:] or [""]:
        how, nl = (variant or "p,4").split(",")
        fn = lambda: pl.train_step(pipe, views, targets, grads, concurrent_views=int(nl), batch_binning=how[0] == "b",
                                   bin_group=int(how[1:] or 4))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        chk = grads.flat.double().abs().sum().item()
        ref = chk if ref is None else ref
        print("variant %-12s step %.3f ms   grad checksum rel diff %.2e" % (variant or "lanes", e0.elapsed_time(e1) / 10,
                                                                            abs(chk - ref) / ref), flush=True)


if __name__ == "__main__":
    main()
