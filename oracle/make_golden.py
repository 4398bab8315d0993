"""Generate tests/golden/*.npz by running the REFERENCE's own CPU Python path.

Runs only in the authoring container (needs /root/reference). It imports the reference's
forward_pass.py / rasterization.py unmodified; the two third-party modules that are not installed
here (numpy-quaternion, pyopencl -- requirements.txt:2,6) are satisfied by inert local stand-ins
exactly as SURVEY.md Appendix A prescribes (pyopencl is never called on the CPU path; the
quaternion stand-in restates as_quat_array/as_rotation_matrix, [w,x,y,z], divide by squared norm).
Nothing of the reference is copied: the fixtures hold inputs and outputs only.

    python oracle/make_golden.py            # writes tests/golden/scene_*.npz

Scenes:
  s0  N=64   256x256 deg3 identity camera, seed 0   (the survey's smoke values)
  s1  N=1500 256x256 deg3 rotated+translated camera
  s2  N=300  256x256 deg0 with an off-screen gaussian (Q10), a z<=2 gaussian (Q6), NaN-free
  s3  N=400  203x117 (not a multiple of 16: Q9) deg2, non-zero background
  s4  N=200  256x256 deg1 equal-depth ties (Q12) -- forward_pass outputs only (the reference's
      rasterize() takes a depth-keyed dict and drops ties, B11)
All float inputs are float32-representable so the fp32 device path sees identical values.
"""
import io
import os
import sys
import types
import contextlib
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    cl = types.ModuleType("pyopencl")
    alg = types.ModuleType("pyopencl.algorithm")
    alg.RadixSort = object
    cl.algorithm = alg
    sys.modules["pyopencl"] = cl
    sys.modules["pyopencl.algorithm"] = alg
    q = types.ModuleType("quaternion")
    q.as_quat_array = lambda a: np.asarray(a, dtype=np.float64)

    def as_rotation_matrix(qa):
        w, x, y, z = qa
        s = 2.0 / (w * w + x * x + y * y + z * z)
        return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                         [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                         [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])
    q.as_rotation_matrix = as_rotation_matrix
    sys.modules["quaternion"] = q
    import forward_pass
    import rasterization
    return forward_pass, rasterization


class Cam:
    def __init__(self, fx, fy, w, h):
        self.params = [fx, fy, w / 2.0, h / 2.0]
        self.width = w
        self.height = h


class Gs:
    def __init__(self, center, scaling, rotation, sh, opacity, degrees):
        self.center, self.scaling, self.rotation = center, scaling, rotation
        self.spherical_harmonics, self.opacity, self.degrees = sh, opacity, degrees


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rot_yxz(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


def make_scene(name):
    if name == "s0":   # SURVEY.md Appendix A smoke scene
        rng = np.random.default_rng(0)
        N = 64
        c = np.concatenate([rng.uniform(-2, 2, (N, 2)), rng.uniform(3, 8, (N, 1))], axis=1)
        sc = rng.uniform(-1, 1, (N, 3)); q = rng.normal(size=(N, 4))
        sh = rng.normal(0, .3, (N, 3, 16)); op = rng.uniform(.1, .9, N)
        return dict(center=c, scaling=sc, rotation=q, sh=sh, opacity=op, degrees=3, fx=300., fy=300.,
                    cam_w=256, cam_h=256, R=np.eye(3), t=np.zeros(3), result_size=[256, 256],
                    bg=np.zeros(3), f32_inputs=False)
    rng = np.random.default_rng({"s1": 11, "s2": 12, "s3": 13, "s4": 14}[name])
    if name == "s1":
        N = 1500
        c = np.concatenate([rng.uniform(-3, 3, (N, 2)), rng.uniform(3, 9, (N, 1))], axis=1)
        sc = rng.uniform(0.5, 2.6, (N, 3)); q = rng.normal(size=(N, 4))
        sh = rng.normal(0, .3, (N, 3, 16)); op = rng.uniform(.05, .95, N)
        return dict(center=c, scaling=sc, rotation=q, sh=sh, opacity=op, degrees=3, fx=310., fy=295.,
                    cam_w=256, cam_h=256, R=rot_yxz(.15, -.1, .05), t=np.array([.2, -.1, .8]),
                    result_size=[256, 256], bg=np.zeros(3), f32_inputs=True)
    if name == "s2":
        N = 300
        c = np.concatenate([rng.uniform(-2.5, 2.5, (N, 2)), rng.uniform(2.5, 7, (N, 1))], axis=1)
        c[0] = [40.0, 0.0, 5.0]      # far off-screen to the right (Q10)
        c[1] = [0.1, -0.2, 1.5]      # z <= 2: near-culled (Q6)
        c[2] = [0.0, -30.0, 4.0]     # off-screen above
        c[3] = [0.3, 0.3, 2.0]       # exactly z == 2: culled
        sc = rng.uniform(1.0, 2.8, (N, 3)); q = rng.normal(size=(N, 4))
        sh = rng.normal(0, .5, (N, 3, 16)); op = rng.uniform(.05, .99, N)
        return dict(center=c, scaling=sc, rotation=q, sh=sh, opacity=op, degrees=0, fx=280., fy=280.,
                    cam_w=256, cam_h=256, R=np.eye(3), t=np.zeros(3), result_size=[256, 256],
                    bg=np.zeros(3), f32_inputs=True)
    if name == "s3":
        N = 400
        c = np.concatenate([rng.uniform(-3, 3, (N, 1)), rng.uniform(-1.8, 1.8, (N, 1)),
                            rng.uniform(3, 8, (N, 1))], axis=1)
        sc = rng.uniform(0.2, 2.2, (N, 3)); q = rng.normal(size=(N, 4))
        sh = rng.normal(0, .3, (N, 3, 16)); op = rng.uniform(.05, .95, N)
        return dict(center=c, scaling=sc, rotation=q, sh=sh, opacity=op, degrees=2, fx=240., fy=250.,
                    cam_w=203, cam_h=117, R=rot_yxz(-.05, .04, 0.0), t=np.array([0., 0., .3]),
                    result_size=[203, 117], bg=np.array([.25, .5, .125]), f32_inputs=True)
    if name == "s4":
        N = 200
        c = np.concatenate([rng.uniform(-1.5, 1.5, (N, 2)), np.zeros((N, 1))], axis=1)
        c[:, 2] = np.repeat(np.array([3.0, 3.5, 4.25, 6.0]), N // 4)   # 4 depth planes: ties
        sc = rng.uniform(1.5, 2.8, (N, 3)); q = rng.normal(size=(N, 4))
        sh = rng.normal(0, .3, (N, 3, 16)); op = rng.uniform(.05, .95, N)
        return dict(center=c, scaling=sc, rotation=q, sh=sh, opacity=op, degrees=1, fx=300., fy=300.,
                    cam_w=256, cam_h=256, R=np.eye(3), t=np.zeros(3), result_size=[256, 256],
                    bg=np.zeros(3), f32_inputs=True)
    raise KeyError(name)


def _collisions(depths, tt):
    """Gaussians whose integer depth collides with another's inside some tile (B11)."""
    seen = {}
    bad = set()
    for g in range(len(depths)):
        if depths[g] == 0:
            continue
        for tx in range(tt[g, 0, 0], tt[g, 1, 0]):
            for ty in range(tt[g, 0, 1], tt[g, 1, 1]):
                k = (tx, ty, int(depths[g]))
                if k in seen:
                    bad.add(g)
                seen[k] = g
    return sorted(bad)


def run_scene(fp, rz, name, with_image=True, tile_size=16):
    s = make_scene(name)
    if s["f32_inputs"]:
        for k in ("center", "scaling", "rotation", "sh", "opacity", "t", "bg"):
            s[k] = f32(s[k])
    cam = Cam(s["fx"], s["fy"], s["cam_w"], s["cam_h"])
    sink = io.StringIO()
    names = ["centers", "depths", "colors", "conics", "clamped", "tiles_touched", "radii",
             "Ws", "Ms", "Ts", "cov2d", "cov3d", "dirs"]
    for attempt in range(50):
        gs = Gs(s["center"], s["scaling"], s["rotation"], s["sh"], s["opacity"], s["degrees"])
        with contextlib.redirect_stdout(sink):
            outs = fp.forward_pass(None, None, None, cam, s["R"], s["t"], gs,
                                   result_size=s["result_size"], tile_size=tile_size, training=True)
        if not with_image:
            break
        # rasterize() wants mapped_keys[tx][ty] = {depth: gaussian} (rasterization.py:167), so a
        # golden scene needs unique integer depths per tile (B11): nudge colliding gaussians in z
        bad = _collisions(outs[1], outs[5])
        if not bad:
            break
        c = s["center"].copy()
        c[bad, 2] += 0.0037 * (1 + attempt)
        s["center"] = f32(c) if s["f32_inputs"] else c
    else:
        raise RuntimeError(f"{name}: could not remove depth collisions")
    rec = {"in_" + k: np.asarray(v) for k, v in s.items() if k != "f32_inputs"}
    rec["in_tile_size"] = np.asarray(tile_size)
    for n, o in zip(names, outs):
        rec["fp_" + n] = np.asarray(o)
    if with_image:
        centers, depths, colors, conics = outs[0], outs[1], outs[2], outs[3]
        tt = outs[5]
        import math
        tx_n = math.ceil(s["result_size"][0] / tile_size)
        ty_n = math.ceil(s["result_size"][1] / tile_size)
        mk = [[dict() for _ in range(ty_n)] for _ in range(tx_n)]
        for g in range(len(centers)):
            if depths[g] == 0:
                continue
            for tx in range(tt[g, 0, 0], tt[g, 1, 0]):
                for ty in range(tt[g, 0, 1], tt[g, 1, 1]):
                    assert depths[g] not in mk[tx][ty], f"{name}: depth collision in tile {tx},{ty}"
                    mk[tx][ty][int(depths[g])] = g
        with contextlib.redirect_stdout(sink):
            img = rz.rasterize(centers, colors, s["opacity"], conics, mk, training=False,
                               result_size=s["result_size"], tile_size=tile_size,
                               background_color=s["bg"])
        rec["img"] = np.asarray(img)
    return rec


def main():
    fp, rz = import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, with_image in (("s0", True), ("s1", True), ("s2", True), ("s3", True), ("s4", False)):
        rec = run_scene(fp, rz, name, with_image)
        path = os.path.join(OUT, f"scene_{name}.npz")
        np.savez_compressed(path, **rec)
        msg = f"{name}: N={len(rec['in_center'])} visible={int((rec['fp_radii'] > 0).sum())}"
        if with_image:
            msg += f" img max={rec['img'].max()!r} mean={rec['img'].mean()!r}"
        print(msg, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    main()
