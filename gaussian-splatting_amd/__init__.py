"""MI355X-native 3D Gaussian Splatting rasteriser: drop-in for the hot-path functions of
AubreyFeldker/gaussian-splatting (forward_pass / match_gaus_to_tiles / gpu_rasterize / c_rasterize
/ rasterize / backward_pass / setup_gpu), backed by hand-written gfx950 HIP kernels behind the C
ABI in include/gsplat.h.  The directory name is not a Python identifier: import it through the
`gsplat_amd` alias module at the repo root (or importlib.import_module("gaussian-splatting_amd"))."""
from ._lib import GsError, LIB_PATH, SIGNATURES, load as load_library   # noqa: F401
from .api import (DeviceContext, GaussianSet, KernelLibrary, TileBinning, backward_pass, c_rasterize,   # noqa: F401
                  forward_pass, gpu_rasterize, match_gaus_to_tiles, rasterize, setup_gpu, view_stats)
from .pipeline import (FlatGrads, flat_offsets, gaussian_chunks, Optimizer, REFERENCE_LEARNING_RATES, SplatPipeline, View, shard_views,   # noqa: F401
                       all_reduce_sum, auto_lanes, grad_all_reduce_alone, train_step)
from .io import (gaussians_from_point_cloud, knn_distances, load_data, load_gaussians, qvec2rotmat,   # noqa: F401
                 quat_to_rotmat, read_model, read_ply_vertices, save_gaussians, save_ply)
from .functions import (focal_to_fov, get_jacobian_matrix, get_pixel, get_projection_matrix, get_view_matrix,   # noqa: F401
                        inv_sigmoid, sigmoid)
from .densify import Densifier   # noqa: F401
from .train import train_model   # noqa: F401
