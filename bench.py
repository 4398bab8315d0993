#!/usr/bin/env python3
"""Benchmark of the hot path: forward+backward training step over an 8-view batch per GPU of
1M synthetic gaussians at 1920x1080 (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  Weak scaling: every rank renders 8 views per step (global batch 8*N views,
sharded round-robin over the ranks; --views-per-rank 0 --views 8 gives the fixed 8-view global batch
instead); per-gaussian gradients [N*59] fp32 are summed with one RCCL all-reduce inside the timed
region.  A step = for every local view preprocess -> binning/sort -> blend forward -> MSE loss
gradient -> blend backward, then ONE batched parameter backward over the local views, then the
all-reduce.
Rank 0 prints ONE JSON line (contract in the task statement).  Synthetic inputs follow SURVEY.md
section 8(d), seed 20240601; they are resident in HBM before the timed region starts.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views-per-rank", type=int, default=8, help="views each rank renders per step (weak scaling)")
    ap.add_argument("--views", type=int, default=0, help="fixed GLOBAL batch instead (strong scaling; needs --views-per-rank 0)")
    ap.add_argument("--log-scale", type=float, nargs=2, default=(1.0, 3.0))
    ap.add_argument("--cpu-views", type=int, default=32, help="frames rendered by the CPU baseline (about 0.4 s each on 16 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-timing", choices=["dominant", "all", "none"], default="dominant",
                    help="which kernels get HIP events inside the timed region")
    ap.add_argument("--k8-early-lanes", type=int, default=None,
                    help="lanes whose views get an early parameter-backward launch (default: half of them)")
    ap.add_argument("--concurrent-views", type=int, default=4, help="local views in flight on separate streams")
    ap.add_argument("--forward-only", action="store_true", help="configs[1]: single-view forward render")
    return ap.parse_args()


def make_scene(N, W, H, log_scale, seed=20240601):
    """x,y ~ U(-1,1)*z*(W or H)/(2f), z ~ U(3,30), log-scales ~ U(lo,hi), unit
    quaternions, opacity = sigmoid(N(0,1.5)) stored as logits, SH ~ N(0,0.3) degree 3."""
    import types
    rng = np.random.default_rng(seed)
    fx = 1100.0 * W / 1920.0
    z = rng.uniform(3, 30, N)
    x = rng.uniform(-1, 1, N) * z * W / (2 * fx)
    y = rng.uniform(-1, 1, N) * z * H / (2 * fx)
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    g = types.SimpleNamespace(center=f32(np.stack([x, y, z], axis=1)),
                              scaling=f32(rng.uniform(log_scale[0], log_scale[1], (N, 3))), rotation=f32(q),
                              spherical_harmonics=f32(rng.normal(0, .3, (N, 3, 16))),
                              opacity=f32(rng.normal(0, 1.5, N)), degrees=3)
    cam = types.SimpleNamespace(params=[fx, fx, W / 2.0, H / 2.0], width=W, height=H)
    return g, cam


def make_views(gs, cam, n_views):
    """Small yaw/pitch rotations about the origin: every gaussian stays in front of the near cull
    in every view, so each view carries the same 1M-gaussian workload (the reference takes the 2D
    centre from world coordinates, quirk Q1; the view changes depth order and covariances)."""
    views = []
    for k in range(n_views):
        yaw = math.radians((k - (n_views - 1) / 2.0) * 1.5)
        pitch = math.radians(((k * 3) % n_views - (n_views - 1) / 2.0) * 0.6)
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        views.append(gs.View(cam, Ry @ Rx, np.zeros(3)))
    return views


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import gsplat_amd as gs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal knobs (a 1-GPU box cannot host two RCCL ranks): GS_BENCH_BACKEND=gloo and
    # GS_BENCH_SHARE_GPU=1 run every rank on cuda:0 with host-staged collectives
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    if os.environ.get("GS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # a collective that does not complete within the timeout ends this rank with a non-zero exit (RCCL: the
        # process group's watchdog aborts the communicator; gloo: the call raises) instead of hanging the job
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("GS_BENCH_COLLECTIVE_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        else:
            dist.init_process_group(backend=backend, timeout=limit)
    dev = torch.device("cuda", local_rank)

    W, H, N = args.width, args.height, args.gaussians
    g, cam = make_scene(N, W, H, args.log_scale)
    ctx, queue, program = gs.setup_gpu(local_rank)
    pipe = gs.SplatPipeline(g, [W, H], opacity_is_logit=True, ctx=ctx)
    weak = args.views_per_rank > 0
    n_views = 1 if args.forward_only else (args.views_per_rank * world if weak else args.views)
    views = make_views(gs, cam, n_views)
    grads = gs.FlatGrads(N, dev)
    mine = gs.shard_views(n_views, rank, world)

    # targets: a fixed pseudo-image per view (the loss only has to produce a dense dL/dimage)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    base_target = torch.rand(W, H, 3, generator=gen).to(dev)
    targets = [base_target for _ in range(n_views)]

    def step():
        if args.forward_only:
            image, _ = pipe.forward(views[0])
            return image
        return gs.train_step(pipe, views, targets, grads, concurrent_views=args.concurrent_views,
                             k8_early_lanes=args.k8_early_lanes)

    trace = os.environ.get("GS_BENCH_TRACE") == "1"

    def mark(what):
        if trace:
            print(f"[rank {rank}] {what}", file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    mark("scene ready")
    for _ in range(args.warmup):
        step()
        mark("warmup step done")
    barrier()
    mark("barrier passed")
    ctxs = pipe.contexts()      # the warm-up steps created every lane's contexts
    # HIP events around every launch cost 0.3 ms of host time per step; by default only the roofline
    # kernel is timed inside the timed region (--kernel-timing all: every kernel, for the `kernels` table)
    names = [ctx.lib.gs_profile_kernel_name(i).decode() for i in range(ctx.lib.gs_profile_kernel_count())]
    # the candidates for "dominant": blend backward (forward when --forward-only) and the tile sort
    few = sum(1 << names.index(k) for k in (("blend_forward",) if args.forward_only else ("blend_backward",)) + ("bin_sort",)) if names else 0
    for c in ctxs:
        mask = 0x7FFFFFFF if args.kernel_timing == "all" else few
        c.set_option("profile_mask", mask)
        c.profile(args.kernel_timing != "none")
        c.read_profile()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {}
    for c in ctxs:
        for k, (ms, cnt) in c.read_profile().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + ms, a[1] + cnt)
        c.profile(False)
        c.set_option("profile_mask", 0x7FFFFFFF)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    mpix = n_views * W * H / 1e6
    value = mpix / (ms_per_step / 1e3)

    # ---- workload statistics on this rank's first view (outside the timed region)
    image, st = pipe.forward(views[mine[0] if mine else 0])
    torch.cuda.synchronize()
    D = st["binning"].D
    ranges = st["binning"].ranges
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.int64)
    tiles_x, tiles_y = st["binning"].tiles_x, st["binning"].tiles_y
    nc = (st["n_contrib"] & 0x7FFFFFFF).to(torch.int64)
    pad = torch.zeros(tiles_x * 16, tiles_y * 16, dtype=torch.int64, device=dev)
    pad[:W, :H] = nc
    per_tile_visited = pad.view(tiles_x, 16, tiles_y, 16).amax(dim=(1, 3)).reshape(-1)
    d_star = int(per_tile_visited.sum().item())           # duplicates visited before early-out
    pairs = int((per_tile_visited * 256).sum().item())
    n_visible = int((st["radii"] > 0).sum().item())
    max_len = int(lens.max().item())

    # ---- roofline of the dominant kernel (largest share of the timed region)
    P = W * H
    alg_bytes = {   # SURVEY.md 8(d) per-launch algorithmic bytes
        "preprocess": N * (236 + 56),
        # tile sort: 2 passes x D x (4 histogram read + 8 read + 8 write); depth sort likewise over N
        "bin_sort": 2 * D * 20,
        "bin_depth_sort": 2 * N * 20,
        "bin_duplicate": N * 28 + D * 8,
        "blend_forward": d_star * 40 + P * 20,
        "blend_backward": d_star * 40 + P * 20 + d_star * 36,
        "param_backward": N * (236 + 36) + N * 236,
    }
    timed = {k: v for k, v in prof.items() if v[1] > 0}
    dom = max(timed, key=lambda k: timed[k][0]) if timed else None
    roofline = None
    kernels = {k: {"ms_per_launch": round(v[0] / v[1], 4), "launches": int(v[1])} for k, v in timed.items()}
    if dom:
        dur_s = timed[dom][0] / timed[dom][1] / 1e3
        ab = alg_bytes.get(dom)
        achieved = ab / dur_s / 1e9 if ab else None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2) if achieved else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None, "traffic": None,
                    "avg_launch_ms": round(dur_s * 1e3, 4), "algorithmic_bytes_per_launch": int(ab) if ab else None,
                    "note": "blend kernels are FP32-VALU bound (no dense contraction, no MFMA); "
                            "valu_* fields price the same launch against the vector-FP32 peak"}
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), collected
        # separately with rocprofv3 --pmc on the same scene (tools/one_view.py, tools/pmc_summary.py)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            if N == 1_000_000 and (W, H) == (1920, 1080) and dom in pmc and "hbm_traffic_bytes" in pmc[dom]:
                roofline["traffic"] = int(pmc[dom]["hbm_traffic_bytes"])
                roofline["traffic_source"] = "profiles/r01_pmc_summary.json"
                # avg_launch_ms above is measured while four views share the GPU; the same kernel alone:
                roofline["single_stream_launch_ms"] = round(pmc[dom]["avg_duration_us_under_pmc"] / 1e3, 4)
        except (OSError, ValueError):
            pass
        if dom in ("blend_forward", "blend_backward"):
            flop_per_pair = 26 if dom == "blend_forward" else 70
            tf = pairs * flop_per_pair / dur_s / 1e12
            roofline.update({"valu_pair_evals_per_s": round(pairs / dur_s / 1e9, 3),
                             "valu_unit": "Gpairs/s", "valu_tflops_est": round(tf, 2),
                             "valu_peak_tflops": FP32_VALU_PEAK_TFLOPS,
                             "valu_frac_est": round(tf / FP32_VALU_PEAK_TFLOPS, 4)})

    # ---- CPU baseline: restatement of rasterize.c on the host cores (rank 0, N=1 only), bounded sample
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cref
        # the reference's own tile lists (its CPU path knows nothing of the culled ones), ~10-15 s of host work
        exact = gs.SplatPipeline(g, [W, H], opacity_is_logit=True, ctx=ctx, cull_tiles=False)
        n_cpu_views, cpu_s, used, d_exact = args.cpu_views, 0.0, 1, 0
        inputs = []
        for v in range(min(n_cpu_views, n_views)):
            _, sv = exact.forward(views[v])
            torch.cuda.synchronize()
            inputs.append([sv[k].cpu().numpy() for k in ("centers", "colors", "opac", "conics")] +
                          [sv["binning"].vals.cpu().numpy(), sv["binning"].ranges.cpu().numpy()])
            d_exact = sv["binning"].D
        for v in range(n_cpu_views):
            a = inputs[v % len(inputs)]
            t1 = time.perf_counter()
            _, _, _, _, used = cref.rasterize_frame(a[0], a[1], a[2], a[3], a[4], a[5], [W, H], (0, 0, 0),
                                                    pixel_offset=0.5, margin=0.0)
            cpu_s += time.perf_counter() - t1
        cpu_baseline = {"value": round(n_cpu_views * P / 1e6 / cpu_s, 3), "unit": "Mpix/s (forward blend only)",
                        "cores": int(used), "kind": "port",
                        "sample": f"{n_cpu_views} views {W}x{H}, forward blend only (the reference's rasterize.c has no "
                                  f"backward), same scene, the reference's tile lists (D={d_exact}), fp64 POSIX restatement of rasterize.c, "
                                  f"OpenMP over tiles, {cpu_s:.2f} s wall on {int(used)} threads; host has "
                                  f"{os.cpu_count()} logical CPUs"}

    if rank == 0:
        out = {
            "metric": "Mpix/s fwd+bwd at 1M gaussians 1080p" if not args.forward_only else "Mpix/s forward at 1M gaussians 1080p",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[2]: fwd+bwd training step, %d gaussians, %d views per GPU per step at %dx%d "
                                    "(global batch %d views sharded over %d rank(s)), RCCL all-reduce of [N*59] fp32 grads"
                                    % (N, len(mine), W, H, n_views, world)) if not args.forward_only else
                                   "configs[1]: single %dx%d view forward, %d gaussians" % (W, H, N),
                       "gaussians": N, "views_per_step": n_views, "width": W, "height": H, "tile": 16,
                       "log_scale_range": list(args.log_scale), "seed": 20240601,
                       "n_visible": n_visible, "duplicates_D": int(D), "visited_D_star": d_star,
                       "max_tile_list": max_len, "pixels_per_thread": 4,
                       "concurrent_views_per_rank": args.concurrent_views},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
