"""Diagnostic build + run: per-wave start/end stamps of the blend kernels (who finishes when, on
which SIMD).  Builds tools/libgsplat_stamp.so with -DGS_STAMPS (the product library has no stamps).
    python tools/stamp_blend.py build      (CPU container)
    python tools/stamp_blend.py run        (GPU box; prints a summary)"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "gaussian-splatting_amd")
LIB = os.path.join(ROOT, "tools", "libgsplat_stamp.so")


def build():
    srcs = ["ctx.hip", "preprocess.hip", "binning.hip", "bin_batch.hip", "blend.hip", "param_bwd.hip", "tile_shim.hip", "train.hip", "stats.hip"]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DGS_STAMPS",
           "-fno-slp-vectorize", "-shared", "-o", LIB] + [os.path.join(PKG, "csrc", s) for s in srcs]
    subprocess.check_call(cmd)


def run():
    import importlib
    _lib = importlib.import_module("gaussian-splatting_amd._lib")
    _lib.LIB_PATH = LIB
    import torch
    import gsplat_amd as gs
    from bench import make_scene, make_views
    N, W, H = 1_000_000, 1920, 1080
    g, cam = make_scene(N, W, H, (1.0, 3.0))
    ctx, _, _ = gs.setup_gpu(0)
    lib = ctx.lib
    lib.gs_debug_read_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
    views = make_views(gs, cam, 8)
    grads = gs.FlatGrads(N, ctx.device)
    target = torch.rand(W, H, 3, device=ctx.device)
    for v in range(2):
        image, st = pipe.forward(views[v])
        pipe.backward(st, (image - target) * 1e-6, grads)
    torch.cuda.synchronize()
    hist = np.zeros(16, dtype=np.uint64)
    lib.gs_debug_read_valid_hist.argtypes = [ctypes.c_void_p]
    assert lib.gs_debug_read_valid_hist(hist.ctypes.data) == 0
    tot = hist.sum()
    print("K7 reduced entries by valid pixels (2 views):", {f"<{1 << b}": f"{int(hist[b])} ({hist[b] / tot:.1%})" for b in range(1, 10) if hist[b]})
    ntiles = 120 * 68
    for which, name in ((0, "forward"), (1, "backward")):
        buf = np.zeros((ntiles, 4), dtype=np.uint64)
        rc = lib.gs_debug_read_stamps(which, buf.ctypes.data, ntiles)
        assert rc == 0
        live = buf[:, 1] > 0
        t0 = buf[live, 0].astype(np.int64); t1 = buf[live, 1].astype(np.int64)
        work = buf[live, 3].astype(np.int64)
        base = t0.min()
        s, e = (t0 - base) / 100.0, (t1 - base) / 100.0     # us (100 MHz)
        life = e - s
        hw = buf[live, 2]
        xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
        hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
        simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
        unit = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        print(f"{name}: waves {live.sum()}, kernel span {e.max():.0f} us; start p50/p90/max {np.percentile(s,50):.0f}/"
              f"{np.percentile(s,90):.0f}/{s.max():.0f} us; lifetime mean {life.mean():.0f} p50 {np.percentile(life,50):.0f} "
              f"p99 {np.percentile(life,99):.0f} max {life.max():.0f} us; end p50/p90/p99 {np.percentile(e,50):.0f}/"
              f"{np.percentile(e,90):.0f}/{np.percentile(e,99):.0f}")
        print(f"  work (list entries) mean {work.mean():.0f} max {work.max()}, corr(life, work) {np.corrcoef(life, work)[0,1]:.2f}")
        # resident waves over time
        ts = np.linspace(0, e.max(), 13)[:-1]
        print("  resident waves at t:", [(int(t), int(((s <= t) & (e > t)).sum())) for t in ts])
        u, cnt = np.unique(unit, return_counts=True)
        print(f"  distinct CUs used {len(u)}, waves per CU min/mean/max {cnt.min()}/{cnt.mean():.1f}/{cnt.max()}")
        # per-CU busy end
        last_end = np.array([e[unit == x].max() for x in u])
        print(f"  per-CU last end: p10 {np.percentile(last_end,10):.0f} p50 {np.percentile(last_end,50):.0f} max {last_end.max():.0f}")
        tile_idx = np.nonzero(live)[0]
        late = tile_idx[np.argsort(-e)[:8]]
        print("  latest tiles (tx,ty,start,end,work):", [(int(t // 68), int(t % 68), int(s[np.where(tile_idx==t)[0][0]]), int(e[np.where(tile_idx==t)[0][0]]), int(work[np.where(tile_idx==t)[0][0]])) for t in late])


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
