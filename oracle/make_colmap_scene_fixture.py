"""ORACLE tooling (authoring container only): BASELINE.json configs[0] as a fixture -- a small COLMAP scene
(~10 k points, one 256x256 PINHOLE camera, three images) written with the REFERENCE's own writer
(colmap/read_write_model.py:512 write_model) into tests/golden/colmap_scene/, plus expected.npz holding what the
reference's own code computes from it:
  * knn_distances (gaussian.py:59-61) of the point cloud;
  * forward_pass.forward_pass(training=False) for every image (forward_pass.py:25-136): integer outputs in full,
    float outputs for every 8th gaussian;
  * rasterization.rasterize (pure Python, rasterization.py:155-215) for image 1 on a 96x64 crop of tiles (the
    interpreted loop needs a minute per 30 k (gaussian, tile) pairs).
The reference modules are imported unmodified with the stand-ins of SURVEY.md Appendix A for the two third-party
modules that are not installed.  The fixtures are data; no reference source is copied.
tests/test_gpu_config0.py takes the scene through the package's reader, initialiser and kernels."""
import contextlib
import io
import math
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference   # noqa: E402  (also puts /root/reference on sys.path)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "colmap_scene")
N, W, H, FX = 10_000, 256, 256, 300.0


def main():
    fp, rz = import_reference()
    from colmap import read_write_model as rw
    import gaussian as ref_gaussian
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)
    # unique integer depth keys for image 1 (R = I, t = 0): z = 3 + (k + .5) / 1000, k a permutation of 0..N-1
    z = 3.0 + (rng.permutation(N) + 0.5) * 1e-3
    lim = 0.5 * W / FX
    u, v = rng.uniform(-1.05, 1.05, N), rng.uniform(-1.05, 1.05, N)
    # a few clusters on top of the uniform spread, so tiles differ in depth complexity
    k = rng.integers(0, 6, N)
    cu, cv = rng.uniform(-.7, .7, 6), rng.uniform(-.7, .7, 6)
    clustered = rng.random(N) < 0.4
    u = np.where(clustered, cu[k] + rng.normal(0, .08, N), u)
    v = np.where(clustered, cv[k] + rng.normal(0, .08, N), v)
    xyz = np.stack([u * lim * z, v * lim * z, z], axis=1)
    rgb = rng.integers(0, 256, size=(N, 3))
    cameras = {1: rw.Camera(id=1, model="PINHOLE", width=W, height=H, params=np.array([FX, FX, W / 2.0, H / 2.0]))}
    poses = {1: (np.array([1.0, 0, 0, 0]), np.zeros(3)),
             2: (np.array([0.9987502603949663, 0.0, 0.04997916927067833, 0.0]), np.array([0.15, -0.05, 0.4])),
             3: (np.array([0.9950041652780258, 0.07, -0.06, 0.02]), np.array([-0.2, 0.1, 1.0]))}
    images = {}
    for iid, (q, t) in poses.items():
        q = q / np.linalg.norm(q)
        images[iid] = rw.Image(id=iid, qvec=q, tvec=t, camera_id=1, name=f"{iid:06}.jpg",
                               xys=np.zeros((0, 2)), point3D_ids=np.zeros(0, dtype=np.int64))
    points = {}
    for i in range(N):
        points[i + 1] = rw.Point3D(id=i + 1, xyz=xyz[i], rgb=rgb[i], error=0.5, image_ids=np.array([1]),
                                   point2D_idxs=np.array([0]))
    rw.write_model(cameras, images, points, OUT, ext=".bin")
    cams, imgs, pts = rw.read_model(OUT, ext=".bin")

    # gaussians as gaussian.py:9-29 builds them (that constructor is shadowed at HEAD, defect B1), with the SH layout
    # the live forward path needs ([N,3,16], load_model.py:40) and rgb scaled to 0..1
    centers = np.array([pts[p].xyz for p in pts])
    knn = ref_gaussian.knn_distances(centers)
    color = (np.array([pts[p].rgb for p in pts], dtype=np.float64) / 255.0 - .5) / 0.28209479177387814
    sh = np.zeros((N, 3, 16))
    sh[:, :, 0] = color

    class G:
        pass
    g = G()
    g.center = centers
    g.scaling = np.repeat(np.log(np.clip(knn, a_min=.0000001, a_max=None)), 3).reshape(N, 3)
    g.rotation = np.full((N, 4), [0.0, 0.0, 0.0, 1.0])
    g.spherical_harmonics = sh
    g.opacity = np.full(N, .1)
    g.degrees = 0
    # the device path takes fp32 parameters: round them first so both sides see the same inputs
    for name in ("center", "scaling", "rotation", "spherical_harmonics", "opacity"):
        setattr(g, name, np.asarray(getattr(g, name), dtype=np.float32).astype(np.float64))

    exp = {"knn_distances": knn, "n_points": np.array(N)}
    sink = io.StringIO()
    names = ["centers", "depths", "colors", "conics", "tiles_touched", "radii"]
    for iid in sorted(imgs):
        R = rw.qvec2rotmat(imgs[iid].qvec)
        with contextlib.redirect_stdout(sink):
            outs = fp.forward_pass(None, None, None, cams[1], R, imgs[iid].tvec, g, result_size=[W, H], tile_size=16,
                                   training=False)
        for n_, o in zip(names, outs):
            o = np.asarray(o)
            exp[f"img{iid}_{n_}"] = o if n_ in ("depths", "tiles_touched", "radii") else o[::8]
        exp[f"img{iid}_R"] = R
        if iid == 1:
            full = outs
    # pure-Python rasterize of image 1 on the tiles [2,8) x [5,9): a 96 x 64 pixel frame of its own whose
    # gaussians are shifted by the crop origin (the function has no crop argument)
    c, depths, colors, conics, tt, radii = (np.asarray(o) for o in full)
    x0, x1, y0, y1 = 2, 8, 5, 9
    cw, ch = (x1 - x0) * 16, (y1 - y0) * 16
    mk = [[dict() for _ in range(y1 - y0)] for _ in range(x1 - x0)]
    for i in range(N):
        if depths[i] == 0:
            continue
        for tx in range(max(tt[i, 0, 0], x0), min(tt[i, 1, 0], x1)):
            for ty in range(max(tt[i, 0, 1], y0), min(tt[i, 1, 1], y1)):
                assert int(depths[i]) not in mk[tx - x0][ty - y0]
                mk[tx - x0][ty - y0][int(depths[i])] = i
    shifted = c - np.array([x0 * 16.0, y0 * 16.0])
    with contextlib.redirect_stdout(sink):
        img = rz.rasterize(shifted, colors, g.opacity, conics, mk, training=False, result_size=[cw, ch], tile_size=16,
                           background_color=np.zeros(3))
    exp["img1_crop_tiles"] = np.array([x0, x1, y0, y1])
    exp["img1_crop_image"] = np.asarray(img)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    print("wrote", {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})
    print("visible per image", [int((exp[f"img{i}_radii"] > 0).sum()) for i in sorted(imgs)],
          "pairs img1", int(sum(len(d) for row in mk for d in row)), "crop image max", float(img.max()))


if __name__ == "__main__":
    main()
