"""ORACLE tooling (authoring container only): writes a tiny synthetic COLMAP model with the
REFERENCE's own writer (colmap/read_write_model.py:512 write_model) into tests/golden/colmap/ as
.bin and .txt, plus expected.npz with the values that went in.  The fixtures are data; no
reference source is copied.  tests/test_io.py reads them back with the repo's own reader."""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from colmap import read_write_model as rw   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "colmap")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(42)
    cameras = {
        1: rw.Camera(id=1, model="PINHOLE", width=979, height=546, params=np.array([1100.5, 1099.25, 489.5, 273.0])),
        2: rw.Camera(id=2, model="SIMPLE_RADIAL", width=640, height=480, params=np.array([500.0, 320.0, 240.0, 0.0125])),
        7: rw.Camera(id=7, model="OPENCV", width=1920, height=1080,
                     params=np.array([1500.0, 1501.0, 960.0, 540.0, 0.01, -0.02, 0.001, 0.002])),
    }
    images, exp = {}, {}
    for k, iid in enumerate((1, 2, 5)):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        n2d = (0, 3, 5)[k]
        xys = rng.uniform(0, 500, size=(n2d, 2))
        ids = rng.integers(-1, 40, size=n2d).astype(np.int64)
        images[iid] = rw.Image(id=iid, qvec=q, tvec=rng.normal(size=3), camera_id=(1, 2, 7)[k],
                               name=f"{iid:06}.jpg", xys=xys, point3D_ids=ids)
    points = {}
    for pid in (3, 4, 9, 21, 22, 30):
        tl = int(rng.integers(1, 4))
        points[pid] = rw.Point3D(id=pid, xyz=rng.normal(size=3) * 3, rgb=rng.integers(0, 256, size=3),
                                 error=float(rng.uniform(0, 2)), image_ids=rng.integers(1, 6, size=tl),
                                 point2D_idxs=rng.integers(0, 5, size=tl))
    rw.write_model(cameras, images, points, OUT, ext=".bin")
    rw.write_model(cameras, images, points, OUT, ext=".txt")
    # read back with the reference's reader: what a user of the reference would see
    c2, i2, p2 = rw.read_model(OUT, ext=".bin")
    flat = {}
    for cid, c in c2.items():
        flat[f"cam{cid}_params"] = np.asarray(c.params)
        flat[f"cam{cid}_wh"] = np.array([c.width, c.height])
        flat[f"cam{cid}_model"] = np.array(c.model)
    for iid, im in i2.items():
        flat[f"img{iid}_qvec"] = np.asarray(im.qvec)
        flat[f"img{iid}_tvec"] = np.asarray(im.tvec)
        flat[f"img{iid}_cam"] = np.array(im.camera_id)
        flat[f"img{iid}_name"] = np.array(im.name)
        flat[f"img{iid}_xys"] = np.asarray(im.xys).reshape(-1, 2)
        flat[f"img{iid}_p3d"] = np.asarray(im.point3D_ids)
        flat[f"img{iid}_R"] = rw.qvec2rotmat(im.qvec)
    for pid, p in p2.items():
        flat[f"pt{pid}_xyz"] = np.asarray(p.xyz)
        flat[f"pt{pid}_rgb"] = np.asarray(p.rgb)
        flat[f"pt{pid}_err"] = np.array(p.error)
        flat[f"pt{pid}_img"] = np.asarray(p.image_ids)
        flat[f"pt{pid}_idx"] = np.asarray(p.point2D_idxs)
    # the reference's kNN scale initialisation (gaussian.py:59-61), imported with the stand-ins of
    # SURVEY.md Appendix A for the two uninstalled third-party modules
    import types
    cl = types.ModuleType("pyopencl"); alg = types.ModuleType("pyopencl.algorithm")
    alg.RadixSort = object; cl.algorithm = alg
    sys.modules["pyopencl"] = cl; sys.modules["pyopencl.algorithm"] = alg
    sys.modules["quaternion"] = types.ModuleType("quaternion")
    import gaussian
    centers = np.array([p2[k].xyz for k in sorted(p2)])
    flat["knn_centers"] = centers
    flat["knn_distances"] = gaussian.knn_distances(centers)
    flat["camera_ids"] = np.array(sorted(c2))
    flat["image_ids"] = np.array(sorted(i2))
    flat["point_ids"] = np.array(sorted(p2))
    np.savez(os.path.join(OUT, "expected.npz"), **flat)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
