"""ctypes binding of libgsplat_hip.so (include/gsplat.h).  Fails loudly when the library is
missing: there is no CPU fallback in the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_LIB: another build of the same library (tools/variant_bench.py builds A/B variants with extra -D flags)
LIB_PATH = os.environ.get("GSPLAT_LIB") or os.path.join(_HERE, "libgsplat_hip.so")


class GsCamera(ctypes.Structure):
    _fields_ = [("R", ctypes.c_double * 9), ("t", ctypes.c_double * 3),
                ("fx", ctypes.c_double), ("fy", ctypes.c_double),
                ("cam_width", ctypes.c_int32), ("cam_height", ctypes.c_int32),
                ("res_w", ctypes.c_int32), ("res_h", ctypes.c_int32),
                ("tile_size", ctypes.c_int32), ("sh_degree", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


class GsViewBin(ctypes.Structure):
    _fields_ = [("tiles_touched", ctypes.c_void_p), ("depths", ctypes.c_void_p), ("records", ctypes.c_void_p),
                ("vals", ctypes.c_void_p), ("ranges", ctypes.c_void_p), ("D_dev", ctypes.c_void_p)]


class GsViewPrepare(ctypes.Structure):
    _fields_ = [("tiles_touched", ctypes.c_void_p), ("depths", ctypes.c_void_p), ("records", ctypes.c_void_p),
                ("perm", ctypes.c_void_p), ("offsets", ctypes.c_void_p)]


class GsViewBackward(ctypes.Structure):
    _fields_ = [("cam", ctypes.POINTER(GsCamera)), ("radii", ctypes.c_void_p), ("clamped", ctypes.c_void_p),
                ("grad_rows", ctypes.c_void_p)]


class GsViewPreprocess(ctypes.Structure):
    _fields_ = [("cam", ctypes.POINTER(GsCamera))] + [(n, ctypes.c_void_p) for n in
                ("centers", "depths", "colors", "conics", "clamped", "tiles_touched", "radii")]


class GsViewRecords(ctypes.Structure):
    _fields_ = [("cam", ctypes.POINTER(GsCamera))] + [(n, ctypes.c_void_p) for n in
                ("records", "depths", "clamped", "tiles_touched", "radii")]


class GsStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ("n_gaussians", "n_visible", "duplicates", "visited", "max_tile_list", "nonempty_tiles", "tiles",
                 "pixels", "pixels_broke", "bytes_preprocess", "bytes_depth_sort", "bytes_duplicate",
                 "bytes_tile_sort", "bytes_ranges", "bytes_blend_forward", "bytes_blend_backward",
                 "bytes_param_backward")]


class GsError(RuntimeError):
    pass


_P = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_F = ctypes.c_float

# name -> (restype, argtypes); every symbol include/gsplat.h declares
SIGNATURES = {
    "gs_ctx_create": (ctypes.c_int, [ctypes.c_int, _P, ctypes.POINTER(_P)]),
    "gs_ctx_set_stream": (ctypes.c_int, [_P, _P]),
    "gs_ctx_destroy": (ctypes.c_int, [_P]),
    "gs_ctx_info": (_I64, [_P, ctypes.c_char_p]),
    "gs_ctx_reserve": (ctypes.c_int, [_P, _I32, _I64, _I32, _I32, _I32, _I32, _I32]),
    "gs_ctx_set_option": (ctypes.c_int, [_P, ctypes.c_char_p, _I32]),
    "gs_profile_kernel_count": (ctypes.c_int, []),
    "gs_profile_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
    "gs_ctx_profile": (ctypes.c_int, [_P, _I32]),
    "gs_ctx_profile_read": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64)]),
    "gs_last_error": (ctypes.c_char_p, []),
    "gs_version": (ctypes.c_int, []),
    "gs_preprocess": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, ctypes.POINTER(GsCamera)] + [_P] * 13),
    "gs_preprocess_batch": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _I32, ctypes.POINTER(GsViewPreprocess)]),
    "gs_preprocess_batch_records": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _I32, ctypes.POINTER(GsViewRecords)]),
    "gs_ctx_use_records": (ctypes.c_int, [_P, _I32, _P]),
    "gs_bin_count": (ctypes.c_int, [_P, _I32, _P, _P, _I32, _I32, _I32, ctypes.POINTER(_I64)]),
    "gs_bin_fill": (ctypes.c_int, [_P, _I32, _P, _P, _I32, _I32, _I32, _I64, _P, _P, _P]),
    "gs_bin_count_culled": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, ctypes.c_float,
                                           ctypes.POINTER(_I64)]),
    "gs_bin_count_culled_begin": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, ctypes.c_float]),
    "gs_bin_count_begin": (ctypes.c_int, [_P, _I32, _P, _P, _I32, _I32, _I32]),
    "gs_bin_count_end": (ctypes.c_int, [_P, ctypes.POINTER(_I64)]),
    "gs_bin_bounds_batch": (ctypes.c_int, [_P, _I32, _I32, ctypes.POINTER(GsViewBin), _I32, _I32, _I32, ctypes.POINTER(_I64)]),
    "gs_bin_culled_batch": (ctypes.c_int, [_P, _I32, _I32, ctypes.POINTER(GsViewBin), _I32, _I32, _I32, ctypes.c_float, _I32,
                                           _I64]),
    "gs_bin_prepare_begin": (ctypes.c_int, [_P, _I32, _I32, ctypes.POINTER(GsViewPrepare), _I32, _I32, _I32, ctypes.c_float]),
    "gs_bin_prepare_end": (ctypes.c_int, [_P, _I32, _I32, ctypes.POINTER(GsViewPrepare), _I32, _I32, _I32,
                                          ctypes.POINTER(_I64)]),
    "gs_bin_fill_prepared": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, ctypes.c_float, _P, _P,
                                            _P]),
    "gs_bin_fill_culled": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, ctypes.c_float, _I64,
                                          _P, _P, _P]),
    "gs_blend_forward": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(_F * 3),
                                        _I32, _I32, _I32, _F, _P, _P, _P]),
    "gs_blend_forward_loss": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(_F * 3),
                                             _I32, _I32, _I32, _F, _P, _P, _P, _P, _I32, _F, _P, _P]),
    "gs_blend_backward": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(_F * 3),
                                         _I32, _I32, _I32, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gs_blend_backward_noalloc": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(_F * 3),
                                                 _I32, _I32, _I32, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gs_backward_params": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, ctypes.POINTER(GsCamera), _P, _P, _P,
                                          _P, _P, _I32, _P, _P, _P, _P]),
    "gs_blend_backward_rows": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(_F * 3),
                                              _I32, _I32, _I32, _F, _P, _P, _P, _P]),
    "gs_backward_params_batch": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _I32, _I32,
                                                ctypes.POINTER(GsViewBackward), _I32, _P, _P, _P, _P, _P]),
    "gs_view_stats": (ctypes.c_int, [_P, _I32, _P, _I64, _P, _I32, _I32, _I32, _P, ctypes.POINTER(GsStats)]),
    "gs_image_loss": (ctypes.c_int, [_P, _I64, _P, _P, _I32, _F, _I32, _P, _P]),
    "gs_adam_step": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.POINTER(_F * 5),
                                    _F, _F, _F, _I32, _I32, _F]),
    "rasterize_tile": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().gs_last_error()
        raise GsError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
