#!/usr/bin/env python3
"""Benchmark of the hot path: forward+backward training step over an 8-view batch per GPU of
1M synthetic gaussians at 1920x1080 (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` is checked, not trusted: with N > 1 and no torch.distributed environment (WORLD_SIZE unset) this
process starts the N ranks itself (a child `python -m torch.distributed.run ...` of this same file, before any GPU
call) and exits with the child's status; with WORLD_SIZE set and different from N it exits non-zero.  The line
reports `ranks_seen` = what an all-reduce of ones over the process group returned.

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
PMC_SUMMARY = "r03_pmc_summary.json"
# counter summaries of other workloads (tools/pmc_collect.sh with GS_N / GS_W / GS_H / GS_LS* / GS_LISTS): the one whose
# source hash and workload match the run is used
PMC_SUMMARIES = (PMC_SUMMARY, "r03_pmc_summary_culled.json", "r03_pmc_summary_10m4k.json")
# stages whose launch is several kernels: the counters of the stage are the sums over its parts
PMC_PARTS = {"bin_sort": ("tile_hist", "tile_hist2", "tile_scatter")}     # (+ the 256-block scans, not separable by name)
LISTS_NOTE = {
    "exact": "SplatPipeline(cull_tiles=False): the reference's own (tile, depth) lists -- keys, values and ranges "
             "bit-exact against the reference",
    "culled": "SplatPipeline(cull_tiles=True): the exact lists minus the (gaussian, tile) pairs that cannot reach "
              "alpha >= 1/255 in the tile, same order; images bit-identical, gradients equal to float-atomics rounding",
}
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
    ap.add_argument("--lists", choices=["exact", "culled"], default="exact",
                    help="exact (default): the reference's own (tile, depth) lists, keys / values / ranges bit-exact; "
                         "culled: the same lists minus the pairs that cannot change a pixel (same images)")
    ap.add_argument("--no-other-lists", action="store_true", help="skip the timed step on the other list kind")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the single-stream per-kernel timing pass")
    ap.add_argument("--kernel-timing", choices=["dominant", "all", "none"], default="dominant",
                    help="which kernels get HIP events inside the timed region")
    ap.add_argument("--k8-early-lanes", type=int, default=None,
                    help="lanes whose views get an early parameter-backward launch (default: half of them)")
    ap.add_argument("--concurrent-views", type=int, default=None,
                    help="local views in flight on separate streams (default: train_step's own rule, 4 lanes, or 2 "
                         "when a view carries >= 32 M (gaussian, tile) pairs)")
    ap.add_argument("--forward-only", action="store_true", help="configs[1]: single-view forward render")
    ap.add_argument("--fused-loss", action="store_true", help="train_step(fused_loss=True): loss gradient in K6's epilogue")
    ap.add_argument("--prepared-lists", action="store_true",
                    help="train_step(prepared=True): count / depth order / offsets per batch of views on the main stream")
    return ap.parse_args()


def source_sha16():
    """Hash of the kernel sources and the build recipe: ties a PMC summary under profiles/ to the tree it was
    collected from (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "gaussian-splatting_amd", "csrc")
    for name in sorted(os.listdir(src)) + ["../build.py", "../../include/gsplat.h"]:
        with open(os.path.join(src, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


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


def make_views(gs, cam, n_views, group=None, world=1):
    """Small yaw/pitch rotations about the origin: every gaussian stays in front of the near cull
    in every view, so each view carries the same 1M-gaussian workload (the reference takes the 2D
    centre from world coordinates, quirk Q1; the view changes depth order and covariances).
    group (weak scaling: the views per rank; view k belongs to rank k % world, shard_views): the pattern of `group`
    views repeats for every rank with a small
    offset (0.11 deg yaw, 0.07 deg pitch per rank), so that every rank renders distinct views of the SAME workload --
    one pattern spread over 8 x 8 views would reach 47 deg of yaw, where gaussians leave through the near plane and
    others blow up on screen: another workload per rank."""
    group = n_views if not group else group
    views = []
    for k in range(n_views):
        kk, r = ((k // world) % group, k % world) if group != n_views else (k, 0)
        yaw = math.radians((kk - (group - 1) / 2.0) * 1.5 + 0.11 * r)
        pitch = math.radians(((kk * 3) % group - (group - 1) / 2.0) * 0.6 + 0.07 * r)
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        views.append(gs.View(cam, Ry @ Rx, np.zeros(3)))
    return views


def launch_ranks(args):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks (one per GPU) as a child
    torch.distributed.run of this file and return its exit status.  Nothing here touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
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
    pipe = gs.SplatPipeline(g, [W, H], opacity_is_logit=True, ctx=ctx, cull_tiles=args.lists == "culled")
    weak = args.views_per_rank > 0
    n_views = 1 if args.forward_only else (args.views_per_rank * world if weak else args.views)
    views = make_views(gs, cam, n_views, group=args.views_per_rank if weak and not args.forward_only else None,
                       world=world)
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
                             k8_early_lanes=args.k8_early_lanes, prepared=args.prepared_lists, fused_loss=args.fused_loss)

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

    # the step's collectives alone on the idle GPUs (same slices, same order, outside the timed region): how much of a
    # multi-rank step is the gradient exchange -- it starts when the last view's K8 has run and the next step's K1 needs
    # its result, so nearly all of it is exposed (Multi-GPU)
    comm_alone = 0.0
    if world > 1 and not args.forward_only:
        grads.flat.zero_()
        gs.grad_all_reduce_alone(pipe, grads)
        barrier()
        t1 = time.perf_counter()
        for _ in range(max(3, min(args.steps, 10))):
            gs.grad_all_reduce_alone(pipe, grads)
        barrier()
        comm_alone = (time.perf_counter() - t1) / max(3, min(args.steps, 10))

    t = torch.tensor([elapsed, comm_alone], dtype=torch.float64, device=dev)
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)      # what the process group really spans
    elapsed, comm_alone = float(t[0].item()), float(t[1].item())
    ranks_seen = int(round(float(ones.item())))
    if ranks_seen != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the process group spans {ranks_seen} rank(s)")
    ms_per_step = elapsed * 1e3 / args.steps
    mpix = n_views * W * H / 1e6
    value = mpix / (ms_per_step / 1e3)

    # ---- workload statistics of this rank's first view, from the library (gs_view_stats; outside the timed region)
    image, st = pipe.forward(views[mine[0] if mine else 0])
    torch.cuda.synchronize()
    stats = gs.view_stats(ctx, st["radii"], st["binning"], st["n_contrib"])
    D, d_star, n_visible, max_len = stats["duplicates"], stats["visited"], stats["n_visible"], stats["max_tile_list"]
    P = W * H

    # ---- every kernel ALONE on the chip: one stream, HIP events of the library around each launch.  These are
    # the durations a rocprofv3 --kernel-trace of tools/one_view.py shows (profiles/), and the ones the roofline
    # uses: inside the step a kernel shares the chip with three other lanes and its span says little about it.
    single = {}
    if not args.forward_only and not args.no_single_stream:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_times
        single, _ = kernel_times.measure(pipe, views, targets[0], min(4, len(views)), repeats=2)
    alg_bytes = {   # SURVEY.md 8(d) algorithmic bytes per launch, as gs_view_stats prices them for these sizes
        "preprocess": stats["bytes_preprocess"], "bin_sort": stats["bytes_tile_sort"],
        "bin_depth_sort": stats["bytes_depth_sort"], "bin_duplicate": stats["bytes_duplicate"],
        "bin_ranges": stats["bytes_ranges"], "blend_forward": stats["bytes_blend_forward"],
        "blend_backward": stats["bytes_blend_backward"],
    }
    timed = {k: v for k, v in prof.items() if v[1] > 0}
    kernels = {k: {"ms_per_launch_in_step": round(v[0] / v[1], 4), "launches": int(v[1])} for k, v in timed.items()}
    for k, v in single.items():
        kernels.setdefault(k, {})["ms_per_launch_alone"] = round(v["ms_per_launch"], 4)
    # dominant kernel: the largest share of a view's single-stream time (blend backward; forward when --forward-only)
    if single:
        dom = max((k for k in single if k in alg_bytes), key=lambda k: single[k]["ms_per_launch"] * single[k]["launches"])
    else:
        dom = max(timed, key=lambda k: timed[k][0]) if timed else None
    # HBM bytes and instruction counts per launch from the PMC passes of THIS source tree on THIS workload
    # (tools/pmc_collect.sh, tools/pmc_summary.py): refused when the kernels have changed since, or for another one
    pmc, pmc_why, pmc_file, reasons = None, None, None, []
    for name in PMC_SUMMARIES:
        try:
            cand = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        meta = cand.get("_meta", {})
        if meta.get("source_sha16") != source_sha16():
            reasons.append(f"{name}: another source tree")
        elif (meta.get("gaussians", 1_000_000), meta.get("width", 1920), meta.get("height", 1080)) != (N, W, H) or \
                tuple(meta.get("log_scale", (1.0, 3.0))) != tuple(args.log_scale) or meta.get("lists", "culled") != args.lists:
            reasons.append(f"{name}: another workload")
        else:
            pmc, pmc_file = cand, name
            break
    if pmc is None:
        pmc_why = "none: no counter summary under profiles/ for this source tree and workload (" + \
                  ("; ".join(reasons) or "no file") + ")"

    def stage_record(k):
        """roofline record of one stage: algorithmic bytes / its duration ALONE on the chip against HBM peak, plus
        what the counters say bounds it."""
        dur_ms = single[k]["ms_per_launch"] if k in single else timed[k][0] / timed[k][1]
        ab = alg_bytes.get(k)
        achieved = ab / (dur_ms / 1e3) / 1e9 if ab else None
        r = {"kernel": k, "bound": "hbm", "achieved": round(achieved, 2) if achieved else None,
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
             "traffic": None, "avg_launch_ms": round(dur_ms, 4),
             "avg_launch_source": "HIP events around the launch, kernel alone on the chip (one stream)"
             if k in single else "HIP events around the launch inside the timed region",
             "algorithmic_bytes_per_launch": int(ab) if ab else None}
        if k in timed:
            r["in_step_launch_ms"] = round(timed[k][0] / timed[k][1], 4)
        e = None
        if pmc is not None:
            # the tile sort is three kernels per pass: its counters are the sums over them
            parts = [pmc[n] for n in PMC_PARTS.get(k, (k,)) if n in pmc]
            if parts:
                e = {}
                for q in parts:
                    w = q.get("launches_per_view", 1.0) if k in PMC_PARTS else 1.0     # e.g. two passes per sort
                    for name in ("hbm_traffic_bytes", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"):
                        if name in q:
                            e[name] = e.get(name, 0.0) + q[name] * w
        if e and "hbm_traffic_bytes" in e:
            r["traffic"] = int(e["hbm_traffic_bytes"])
            r["traffic_source"] = f"profiles/{pmc_file} (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
        elif pmc_why:
            r["traffic_source"] = pmc_why
        if e and e.get("GRBM_GUI_ACTIVE") and "SQ_INSTS_VALU" in e:
            cycles = e["GRBM_GUI_ACTIVE"] / 8.0
            issue = {"valu": round(e["SQ_INSTS_VALU"] * 2.0 / (1024 * cycles), 4)}
            if "SQ_INSTS_SALU" in e:
                issue["salu"] = round(e["SQ_INSTS_SALU"] / (256 * cycles), 4)
            if "SQ_LDS_IDX_ACTIVE" in e:
                issue["lds"] = round(e["SQ_LDS_IDX_ACTIVE"] / (256 * cycles), 4)
            top = max(issue, key=issue.get)
            r["issue"] = {"bound": top, "frac": issue[top], "valu_issue_frac": issue.get("valu"),
                          "salu_issue_frac": issue.get("salu"), "lds_busy_frac": issue.get("lds"),
                          "valu_wave_instructions": int(e["SQ_INSTS_VALU"]),
                          "note": "counter-derived: wave-instructions x 2 cycles / (1024 SIMDs x kernel cycles); scalar "
                                  "instructions / (256 scalar units x cycles); LDS-array active cycles / (256 CUs x cycles)"}
        return r

    roofline = None
    if dom:
        roofline = stage_record(dom)
        roofline["note"] = ("frac is the HBM fraction the metric asks for; the blend kernels are bound by vector-instruction "
                            "issue and wave life, not by HBM or MFMA (no dense contraction): see roofline.issue")
        # the stage that IS bound by HBM and furthest from it: the tile sort
        if "bin_sort" in single and dom != "bin_sort":
            roofline["hbm_stage"] = stage_record("bin_sort")
        # the whole step against the HBM roofline: algorithmic bytes of everything a step launches / step time
        if not args.forward_only and stats["duplicates"] >= 0:
            V = len(mine)
            per_view = (stats["bytes_depth_sort"] + N * 56 + N * 16 + stats["bytes_duplicate"] + stats["bytes_tile_sort"] +
                        stats["bytes_ranges"] + stats["bytes_blend_forward"] + 36 * P + stats["bytes_blend_backward"])
            step_bytes = N * (240 + V * 91) + V * per_view + N * (236 + V * 71 + 236)
            roofline["step_algorithmic_bytes"] = int(step_bytes)
            roofline["step_hbm_frac"] = round(step_bytes / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
            roofline["step_note"] = ("K1 N*(240+V*91) + V views x (depth sort, count, scan, duplicate, tile sort, ranges, K6, "
                                     "loss 36*P, K7; view 0's D and D* taken for every view) + K8 N*(236+V*71+236)")

    # ---- the same step on the other kind of lists
    other_lists = None
    if rank == 0 and world == 1 and not args.forward_only and not args.no_other_lists:
        other_kind = "culled" if args.lists == "exact" else "exact"
        other_pipe = gs.SplatPipeline(g, [W, H], opacity_is_logit=True, ctx=ctx, cull_tiles=other_kind == "culled")
        run = lambda: gs.train_step(other_pipe, views, targets, grads, concurrent_views=args.concurrent_views,
                                    k8_early_lanes=args.k8_early_lanes, prepared=args.prepared_lists, fused_loss=args.fused_loss)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) * 1e3 / args.steps
        _, se = other_pipe.forward(views[0])
        other_lists = {"lists": other_kind, "ms_per_step": round(ms, 3), "value": round(mpix / (ms / 1e3), 2),
                       "unit": "Mpix/s", "duplicates_D": int(se["binning"].D), "note": LISTS_NOTE[other_kind]}
        del other_pipe, se

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
                                  f"OpenMP over tiles, built with gcc -O2 -fopenmp -ffp-contract=off (no -march=native: "
                                  f"oracle/Makefile), {cpu_s:.2f} s wall on {int(used)} threads; host has "
                                  f"{os.cpu_count()} logical CPUs"}
        # the same work on the GPU, for a like-for-like reading (the headline is fwd+bwd of the whole path, this
        # baseline is the forward blend alone): K6 on the reference's lists, alone on the chip
        k6_ms = None
        if args.lists == "exact" and "blend_forward" in single:
            k6_ms = single["blend_forward"]["ms_per_launch"]
        else:
            exact.ctx.profile(True)
            exact.ctx.read_profile()
            for _ in range(3):
                exact.forward(views[0])
            pr = exact.ctx.read_profile().get("blend_forward")
            exact.ctx.profile(False)
            k6_ms = pr[0] / pr[1] if pr and pr[1] else None
        if k6_ms:
            cpu_baseline["gpu_same_work"] = {"what": "forward blend (K6) of one view on the same lists, alone on the chip",
                                             "ms_per_view": round(k6_ms, 4), "value": round(P / 1e6 / (k6_ms / 1e3), 1),
                                             "unit": "Mpix/s (forward blend only)"}

    if rank == 0:
        res = "%dx%d" % (W, H)
        what = "%s gaussians %s" % (("%dM" % (N // 1_000_000)) if N % 1_000_000 == 0 else str(N), "1080p" if (W, H) == (1920, 1080) else res)
        is_cfg2 = (N, W, H) == (1_000_000, 1920, 1080) and tuple(args.log_scale) == (1.0, 3.0)
        if args.forward_only:
            workload = "%ssingle %s view forward, %d gaussians" % ("configs[1]: " if is_cfg2 else "", res, N)
        else:
            workload = ("%sfwd+bwd training step, %d gaussians, %d views per GPU per step at %s (global batch %d views "
                        "sharded over %d rank(s)), RCCL all-reduce of [N*59] fp32 grads"
                        % ("configs[2]: " if is_cfg2 and len(mine) == 8 else "", N, len(mine), res, n_views, world))
        out = {
            "metric": ("Mpix/s fwd+bwd at " if not args.forward_only else "Mpix/s forward at ") + what,
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "gaussians": N, "views_per_step": n_views, "width": W, "height": H, "tile": 16,
                       "log_scale_range": list(args.log_scale), "seed": 20240601,
                       "view_set": "per rank: yaw (j-3.5)*1.5 deg, pitch ((3j mod n)-3.5)*0.6 deg about the origin for its n views j (+0.11 / +0.07 deg per rank): every gaussian "
                                   "stays visible in every view",
                       "n_visible": n_visible, "duplicates_D": int(D), "visited_D_star": d_star,
                       "max_tile_list": max_len, "pixels_per_thread": 4,
                       "concurrent_views_per_rank": args.concurrent_views or gs.auto_lanes(pipe),
                       "lists": args.lists, "lists_note": LISTS_NOTE[args.lists], "ranks_seen": ranks_seen},
            "ranks_seen": ranks_seen,
            "collective": None if world == 1 or args.forward_only else {
                "what": "SUM all-reduce of the [N*59] fp32 gradient buffer, 4 ranges of SH gradients + 1 slice of the rest, "
                        "as train_step issues them; timed alone on idle GPUs after the timed region, max over ranks",
                "backend": backend, "bytes": N * 59 * 4, "alone_ms": round(comm_alone * 1e3, 3),
                "algbw_GBps": round(N * 59 * 4 / comm_alone / 1e9, 1) if comm_alone > 0 else None,
                "step_minus_collective_ms": round(ms_per_step - comm_alone * 1e3, 3)},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "other_lists": other_lists, "kernels": kernels,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
