"""How much of a training step is the blend kernels' own throughput?  With the lists of 8 views of the bench scene
already built, times K6 + loss + K7 alone: (a) one stream, view after view; (b) on n concurrent lanes (the
step's arrangement, nothing else running); and the whole train_step beside them.
    python tools/blend_floor.py        (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsplat_amd as gs
from bench import make_scene, make_views
pl = __import__("importlib").import_module("gaussian-splatting_amd.pipeline")


def main():
    N, W, H = int(os.environ.get("GS_N", 1_000_000)), 1920, 1080
    g, cam = make_scene(N, W, H, (1.0, 3.0))
    ctx, _, _ = gs.setup_gpu(0)
    pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
    views = make_views(gs, cam, 8)
    target = torch.rand(W, H, 3, device=ctx.device)
    pres = pipe.preprocess_batch(views)
    toks = []
    for v in range(8):
        tok = pipe.forward_begin(views[v], pre=pres[v])
        pipe.forward_bin(tok)
        toks.append(tok)
    torch.cuda.synchronize()
    losses = torch.zeros(8, device=ctx.device)

    def blend_only(lanes):
        main = ctx.current_stream()
        for _, s in lanes:
            s.wait_stream(main)
        for v in range(8):
            c, s = lanes[v % len(lanes)]
            with ctx.use_stream(s):
                image, st = pipe.forward_blend(toks[v], ctx=c)
                dL, _ = pipe.image_loss(image, target, "l2", 1.0, losses[v:v + 1], ctx=c)
                st["ctx"] = c
                pipe.backward_rows(st, dL)
        for _, s in lanes:
            main.wait_stream(s)

    def timed(fn, reps=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    one = [(ctx, ctx.current_stream())]
    print("K6 + loss + K7 of 8 views, one stream: %.3f ms" % timed(lambda: blend_only(one)), flush=True)
    for n in (2, 3, 4):
        lanes = pipe.lanes(n - 1) + one if n > 1 else one
        print("K6 + loss + K7 of 8 views, %d lanes:  %.3f ms" % (n, timed(lambda: blend_only(lanes))), flush=True)
    grads = gs.FlatGrads(N, ctx.device)
    targets = [target] * 8
    print("whole train_step (8 views, 4 lanes):  %.3f ms" %
          timed(lambda: pl.train_step(pipe, views, targets, grads, concurrent_views=4)), flush=True)


if __name__ == "__main__":
    main()
