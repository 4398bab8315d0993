"""ORACLE tooling (authoring container only): evaluates the REFERENCE's own helper module (functions.py:4-45,
imported from /root/reference -- it has no third-party dependency beyond NumPy) on seeded inputs and writes
inputs + outputs to tests/golden/functions.npz.  Data only; tests/test_functions.py compares the package's
helpers of the same names against it."""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import functions as ref   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "functions.npz")


def main():
    rng = np.random.default_rng(7)
    n = 24
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(n, 3, 3)
    t = rng.normal(size=(n, 3)) * 3
    focal = rng.uniform(200, 2000, size=(n, 2))
    pixels = rng.integers(64, 4000, size=(n, 2)).astype(np.float64)
    # view-space points: inside, outside and far outside the 1.3 * tan(fov / 2) clamp, negative z included
    centers = rng.normal(size=(n, 3)) * np.array([4.0, 4.0, 1.0]) + np.array([0, 0, 3.0])
    centers[3] = [50.0, -40.0, 2.5]
    centers[4] = [-0.1, 0.2, -1.5]
    ndc = rng.uniform(-1.5, 1.5, size=n)
    xs = np.concatenate([rng.normal(size=n - 4) * 4, [0.0, -30.0, 30.0, 1e-3]])
    ps = rng.uniform(1e-4, 1 - 1e-4, size=n)
    out = dict(R=R, t=t, focal=focal, pixels=pixels, centers=centers, ndc=ndc, xs=xs, ps=ps)
    fov = np.array([[ref.focal_to_fov(focal[i, k], pixels[i, k]) for k in range(2)] for i in range(n)])
    out["fov"] = fov
    out["view"] = np.stack([ref.get_view_matrix(R[i], t[i]) for i in range(n)])
    out["proj"] = np.stack([ref.get_projection_matrix(fov[i, 0], fov[i, 1]) for i in range(n)])
    out["proj_nf"] = np.stack([ref.get_projection_matrix(fov[i, 0], fov[i, 1], z_near=0.2, z_far=37.0) for i in range(n)])
    out["jac"] = np.stack([np.asarray(ref.get_jacobian_matrix(fov[i, 0], fov[i, 1], focal[i, 0], focal[i, 1], centers[i]))
                           for i in range(n)])
    out["pixel"] = np.array([ref.get_pixel(ndc[i], pixels[i, 0]) for i in range(n)])
    out["sigmoid"] = ref.sigmoid(xs)
    out["inv_sigmoid"] = ref.inv_sigmoid(ps)
    np.savez(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
