"""The package's helpers under the reference's names (functions.py:4-45) against values the reference's own
module produced on the same inputs (tests/golden/functions.npz, written by oracle/make_functions_fixture.py).
CPU only; /root/reference is not read."""
import os

import numpy as np

from conftest import ROOT   # noqa: F401  (puts the repo root on sys.path)
import gsplat_amd as gs

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "functions.npz"))
N = len(G["ndc"])


def test_reference_helper_names_are_exported():
    for name in ("sigmoid", "inv_sigmoid", "focal_to_fov", "get_view_matrix", "get_projection_matrix",
                 "get_jacobian_matrix", "get_pixel"):
        assert callable(getattr(gs, name)), name


def test_fov_view_projection_match_the_reference_bit_for_bit():
    for i in range(N):
        fov = [gs.focal_to_fov(G["focal"][i, k], G["pixels"][i, k]) for k in range(2)]
        assert fov == list(G["fov"][i])
        assert np.array_equal(gs.get_view_matrix(G["R"][i], G["t"][i]), G["view"][i])
        assert np.array_equal(gs.get_projection_matrix(fov[0], fov[1]), G["proj"][i])
        assert np.array_equal(gs.get_projection_matrix(fov[0], fov[1], z_near=0.2, z_far=37.0), G["proj_nf"][i])
        assert gs.get_pixel(G["ndc"][i], G["pixels"][i, 0]) == G["pixel"][i]


def test_jacobian_is_the_transposed_np_matrix_of_the_reference():
    for i in range(N):
        j = gs.get_jacobian_matrix(G["fov"][i, 0], G["fov"][i, 1], G["focal"][i, 0], G["focal"][i, 1], G["centers"][i])
        assert isinstance(j, np.matrix) and j.shape == (3, 3)
        assert np.array_equal(np.asarray(j), G["jac"][i])
        assert np.all(np.asarray(j)[:, 2] == 0.0)      # transposed: the zero row became a column (quirk Q5)


def test_activations():
    assert np.array_equal(gs.sigmoid(G["xs"]), G["sigmoid"])
    assert np.array_equal(gs.inv_sigmoid(G["ps"]), G["inv_sigmoid"])
    np.testing.assert_allclose(gs.sigmoid(gs.inv_sigmoid(G["ps"])), G["ps"], rtol=1e-12)
