import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_scene(name):
    d = np.load(os.path.join(GOLDEN, f"scene_{name}.npz"))
    return {k: d[k] for k in d.files}


class Cam:
    def __init__(self, fx, fy, w, h):
        self.params = [float(fx), float(fy), w / 2.0, h / 2.0]
        self.width = int(w)
        self.height = int(h)


def scene_inputs(d):
    """(camera, R, t, gaussians-like namespace, result_size, bg) from a golden record."""
    import types
    cam = Cam(d["in_fx"], d["in_fy"], int(d["in_cam_w"]), int(d["in_cam_h"]))
    g = types.SimpleNamespace(center=d["in_center"], scaling=d["in_scaling"], rotation=d["in_rotation"],
                              spherical_harmonics=d["in_sh"], opacity=d["in_opacity"],
                              degrees=int(d["in_degrees"]))
    return cam, d["in_R"], d["in_t"], g, [int(v) for v in d["in_result_size"]], d["in_bg"]


def synth_scene(N, W, H, seed=20240601, fx=None, log_scale=(1.0, 3.0), degree=3):
    """SURVEY.md section 8(d) synthetic distribution (float32-representable)."""
    import types
    rng = np.random.default_rng(seed)
    fx = float(fx if fx is not None else 1100.0 * W / 1920.0)
    z = rng.uniform(3, 30, N)
    x = rng.uniform(-1, 1, N) * z * W / (2 * fx)
    y = rng.uniform(-1, 1, N) * z * H / (2 * fx)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g = types.SimpleNamespace(
        center=f32(np.stack([x, y, z], axis=1)), scaling=f32(rng.uniform(log_scale[0], log_scale[1], (N, 3))),
        rotation=f32(q), spherical_harmonics=f32(rng.normal(0, .3, (N, 3, 16))),
        opacity=f32(1 / (1 + np.exp(-rng.normal(0, 1.5, N)))), degrees=degree)
    return Cam(fx, fx, W, H), np.eye(3), np.zeros(3), g, [W, H]


def dense_scene(N=2500, W=112, H=96, seed=3):
    """Small opaque, heavily overlapping scene: exercises the transmittance break
    (rasterize.c:118-119) and the .99 alpha clamp (rasterize.c:112)."""
    cam, R, t, g, rs = synth_scene(N, W, H, seed=seed, fx=90.0, log_scale=(2.2, 3.4), degree=2)
    rng = np.random.default_rng(seed + 1)
    g.opacity = np.asarray(rng.uniform(.5, 1.0, N), dtype=np.float32)
    g.center[:, 2] = np.asarray(rng.uniform(2.5, 6.0, N), dtype=np.float32)
    return cam, R, t, g, rs


def as_record(cam, R, t, g, rs, bg=(0., 0., 0.)):
    """Golden-record-shaped dict for a synthetic scene (inputs only)."""
    return dict(in_center=np.float64(g.center), in_scaling=np.float64(g.scaling), in_rotation=np.float64(g.rotation),
                in_sh=np.float64(g.spherical_harmonics), in_opacity=np.float64(g.opacity),
                in_degrees=np.asarray(g.degrees), in_fx=np.asarray(cam.params[0]), in_fy=np.asarray(cam.params[1]),
                in_cam_w=np.asarray(cam.width), in_cam_h=np.asarray(cam.height), in_R=np.asarray(R, dtype=np.float64),
                in_t=np.asarray(t, dtype=np.float64), in_result_size=np.asarray(rs), in_bg=np.asarray(bg, dtype=np.float64),
                in_tile_size=np.asarray(16))


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gsplat_amd
    ctx, queue, program = gsplat_amd.setup_gpu()
    yield ctx
    ctx.close()
