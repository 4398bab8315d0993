"""Host-side math helpers under the reference's names (functions.py:4-45), so that callers written against
the reference (`from functions import *`, train.py:8, forward_pass.py:5) find them in this package too.

The kernels do not call these: K1 / K8 derive the same per-frame constants in C (csrc/precam.h,
gs_make_precam) and the per-gaussian Jacobian on the device.  These are the NumPy forms a caller uses to
prepare cameras or activate opacities; tests/test_functions.py pins them to values computed by the
reference's own module on the same inputs (tests/golden/functions.npz)."""
import math

import numpy as np

__all__ = ["get_view_matrix", "get_projection_matrix", "get_jacobian_matrix", "focal_to_fov", "get_pixel",
           "sigmoid", "inv_sigmoid"]


def get_view_matrix(r, t):
    """functions.py:4-9: the 4x4 [R^T | t; 0 0 0 1] world-to-view matrix."""
    m = np.eye(4)
    m[:3, :3] = np.asarray(r, dtype=np.float64).T
    m[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    return m


def get_projection_matrix(fov_x, fov_y, z_near=.01, z_far=100):
    """functions.py:11-22: perspective matrix with w = z (row 3 = [0, 0, 1, 0])."""
    span = z_far - z_near
    p = np.zeros((4, 4))
    # z_near / (tan * z_near), not 1 / tan: the reference's operation order, so the doubles agree to the last bit
    p[0, 0] = z_near / (math.tan(fov_x / 2) * z_near)
    p[1, 1] = z_near / (math.tan(fov_y / 2) * z_near)
    p[2, 2] = z_far / span
    p[2, 3] = -(z_far * z_near) / span
    p[3, 2] = 1.0
    return p


def get_jacobian_matrix(fov_x, fov_y, focal_x, focal_y, center):
    """functions.py:24-34: Jacobian of the perspective projection at `center` (view space), x/z and y/z clamped
    to 1.3 * tan(fov / 2), returned TRANSPOSED as an np.matrix (quirk Q5: forward_pass.py:80 transposes it
    again together with the camera rotation)."""
    x, y, z = (float(center[0]), float(center[1]), float(center[2]))
    lx, ly = 1.3 * math.tan(fov_x * .5), 1.3 * math.tan(fov_y * .5)
    cx = min(lx, max(-lx, x / z)) * z
    cy = min(ly, max(-ly, y / z)) * z
    j = np.matrix([[focal_x / z, 0.0, -(focal_x * cx) / (z * z)],
                   [0.0, focal_y / z, -(focal_y * cy) / (z * z)],
                   [0.0, 0.0, 0.0]])
    return j.T


def focal_to_fov(focal_len, pixels):
    """functions.py:36-37."""
    return 2.0 * math.atan(pixels / (2.0 * focal_len))


def get_pixel(num, scale):
    """functions.py:38-39: NDC coordinate in [-1, 1] -> pixel coordinate, ((v + 1) * S - 1) / 2."""
    return ((num + 1.0) * scale - 1.0) * .5


def sigmoid(x):
    """functions.py:42-43."""
    return 1 / (1 + np.exp(-x))


def inv_sigmoid(x):
    """functions.py:44-45."""
    return -1 * np.log((1 / x) - 1)
