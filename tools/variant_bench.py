"""A/B harness for kernel variants: builds libgsplat variants with extra -D flags and times the
per-kernel HIP-event durations on the bench scene.
    python tools/variant_bench.py build name1:-DFOO=1 name2:-DBAR=2,-DBAZ   (CPU container)
    python tools/variant_bench.py run name1 name2                              (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "gaussian-splatting_amd")
SRCS = ["ctx.hip", "preprocess.hip", "binning.hip", "bin_batch.hip", "blend.hip", "param_bwd.hip", "tile_shim.hip", "train.hip", "stats.hip"]


def lib_path(name):
    return os.path.join(ROOT, "tools", f"libgsplat_{name}.so")


def build(specs):
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        # same flags as gaussian-splatting_amd/build.py (-ffp-contract=off applies to every file here: timing only)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-slp-vectorize",
               "-shared", "-o", lib_path(name)] + flags + [os.path.join(PKG, "csrc", s) for s in SRCS]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            print(out.decode()[-3000:])
            raise SystemExit(f"build of {name} failed")
        print("built", name)


def run_one(name):
    import importlib
    _lib = importlib.import_module("gaussian-splatting_amd._lib")
    _lib.LIB_PATH = lib_path(name) if name != "product" else _lib.LIB_PATH
    import torch
    import gsplat_amd as gs
    from bench import make_scene, make_views
    N, W, H = int(os.environ.get("GS_N", 1_000_000)), 1920, 1080
    lo = float(os.environ.get("GS_LS_LO", 1.0)); hi = float(os.environ.get("GS_LS_HI", 3.0))
    g, cam = make_scene(N, W, H, (lo, hi))
    ctx, _, _ = gs.setup_gpu(0)
    pipe = gs.SplatPipeline(g, [W, H], ctx=ctx)
    views = make_views(gs, cam, 8)
    grads = gs.FlatGrads(N, ctx.device)
    target = torch.rand(W, H, 3, device=ctx.device)
    ref = None
    for rep in range(3):
        if rep == 1:
            ctx.profile(True); ctx.read_profile()
        states = []
        for v in range(4):
            image, st = pipe.forward(views[v])
            pipe.backward_rows(st, (image - target) * 1e-6)
            states.append(st)
        pipe.backward_params(states, grads)
    prof = ctx.read_profile()
    out = {k: round(v[0] / v[1], 4) for k, v in prof.items() if v[1]}
    print(name, {k: out[k] for k in ("blend_forward", "blend_backward", "blend_pack", "param_backward", "bin_sort", "bin_depth_sort",
                                       "bin_duplicate", "preprocess") if k in out},
          "checksum %.6e %.6e" % (image.double().sum().item(), grads.flat.double().abs().sum().item()), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "run1":
        run_one(sys.argv[2])
    else:
        for n in sys.argv[2:]:
            subprocess.call([sys.executable, os.path.abspath(__file__), "run1", n])
