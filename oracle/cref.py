"""ORACLE: ctypes binding of oracle/libgs_oracle.so (rasterize_ref.c, the POSIX restatement of
rasterize.c).  Test infrastructure / CPU baseline only."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgs_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.gso_rasterize_frame.restype = ctypes.c_int
        L.gso_rasterize_frame.argtypes = [dp, ip, ctypes.c_void_p, ctypes.c_void_p, dp, dp, dp, dp,
                                          ip, ip, dp, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_int]
        _LIB = L
    return _LIB


def rasterize_frame(centers, colors, opacities, conics, vals, ranges, result_size,
                    background=(0, 0, 0), pixel_offset=0.5, margin=0.0, nthreads=0):
    """Returns image[W,H,3] f64, last_contributors[W,H] i32, T_final[W,H] f64, fragile[W,H] u8,
    threads_used."""
    W, H = int(result_size[0]), int(result_size[1])
    c = np.ascontiguousarray(centers, dtype=np.float64)
    col = np.ascontiguousarray(colors, dtype=np.float64)
    op = np.ascontiguousarray(np.asarray(opacities).reshape(-1), dtype=np.float64)
    con = np.ascontiguousarray(conics, dtype=np.float64)
    v = np.ascontiguousarray(vals, dtype=np.int32)
    if v.size == 0:
        v = np.zeros(1, dtype=np.int32)
    r = np.ascontiguousarray(ranges, dtype=np.int32)
    bg = np.ascontiguousarray(background, dtype=np.float64)
    img = np.zeros((W, H, 3)); lc = np.zeros((W, H), dtype=np.int32)
    tf = np.ones((W, H)); fr = np.zeros((W, H), dtype=np.uint8)
    used = lib().gso_rasterize_frame(img, lc, tf.ctypes.data, fr.ctypes.data, c, col, op, con, v, r,
                                     bg, W, H, float(pixel_offset), float(margin), int(nthreads))
    return img, lc, tf, fr, used
