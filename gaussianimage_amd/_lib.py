"""ctypes binding of ``csrc/libgi2d.so`` (the C ABI declared in ``include/gi2d.h``).

The library is the only backend of this package: there is no CPU or PyTorch
fallback.  A missing or unloadable library raises at import of this module.
"""
import ctypes
import os

# PyTorch-ROCm carries its own copy of the HIP runtime (same SONAME as /opt/rocm's).  It must
# be the one already mapped when libgi2d.so is loaded, or the process ends up with two HIP
# runtimes and torch's device pointers and streams mean nothing to this library.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgi2d.so")

GI2D_OK = 0
GI2D_E_INVALID = -1
GI2D_E_STATE = -2
GI2D_E_CAPACITY = -3
GI2D_E_PREV_OVERFLOW = -4

BIN_SYNC = 1
BIN_COUNTED = 2
BIN_NOGROW = 4

OPT_CULL = 1
OPT_RASTER_PPL = 2
OPT_DROP_PENDING = 3
OPT_BWD_SERIAL = 4
OPT_RECORDS = 5
OPT_PROJECT_AGG = 6
OPT_FAIL_SIDE_ALLOCS = 7
PROBE = {"project": 1, "raster_fwd": 2, "raster_bwd": 3, "gather": 4, "fit_raster": 5,
         "fit_tail": 6}

BLOCK = 16
MAX_COLOR_SETS = 3  # GI2D_MAX_COLOR_SETS
MAX_VQ_LAYERS = 8   # GI2D_MAX_VQ_LAYERS
NOALLOC_MAX_TILES = 34816        # GI2D_NOALLOC_MAX_TILES
NOALLOC_MAX_LIST_INTS = 2097152  # GI2D_NOALLOC_MAX_LIST_INTS


class AdanHyper(ctypes.Structure):
    """Mirror of ``gi2d_adan_hyper``."""
    _fields_ = [("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("beta3", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double), ("no_prox", ctypes.c_int),
                ("step", ctypes.c_int)]


class Codec(ctypes.Structure):
    """Mirror of ``gi2d_codec``."""
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("xyz_half", "cov_a_code", "cov_a_scale", "cov_a_beta", "cov_b_code",
                 "cov_b_scale", "cov_b_beta", "color_index", "codebooks")] + \
               [("num_quantizers", ctypes.c_int), ("codebook_size", ctypes.c_int)]


class Conv(ctypes.Structure):
    """Mirror of ``gi2d_conv``."""
    _fields_ = [("pixel_offset", ctypes.c_float), ("center_bias", ctypes.c_float),
                ("alpha_cap", ctypes.c_float), ("alpha_min", ctypes.c_float),
                ("bg_weight", ctypes.c_float)]


class FrameArgs(ctypes.Structure):
    """Mirror of ``gi2d_frame_args``."""
    _fields_ = [("ws", ctypes.c_void_p), ("kind", ctypes.c_int)] + \
               [(n, ctypes.c_void_p) for n in ("means2d", "cov_a", "cov_b", "colors", "opacity",
                                               "background")] + \
               [("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("conv", Conv)] + \
               [(n, ctypes.c_void_p) for n in ("xys", "depths", "radii", "conics", "num_tiles_hit",
                                               "out_img", "raster_begin_event", "raster_end_event",
                                               "stream")]


_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
_convp = ctypes.POINTER(Conv)

# name -> (restype, argtypes); every symbol that include/gi2d.h declares
SIGNATURES = {
    "gi2d_version": (_i, []),
    "gi2d_build_id": (ctypes.c_char_p, []),
    "gi2d_error_string": (ctypes.c_char_p, [_i]),
    "gi2d_default_conv": (None, [_convp]),
    "gi2d_workspace_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "gi2d_workspace_destroy": (_i, [_vp]),
    "gi2d_workspace_reserve": (_i, [_vp, _i64, _i]),
    "gi2d_workspace_set_option": (_i, [_vp, _i, _i]),
    "gi2d_workspace_set_rows": (_i, [_vp, _i, _i]),
    "gi2d_workspace_set_probe": (_i, [_vp, _i, _vp, _vp]),
    "gi2d_workspace_info": (_i, [_vp, ctypes.POINTER(_i64)]),
    "gi2d_project_cholesky_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _convp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp]),
    "gi2d_project_rs_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _convp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp]),
    "gi2d_project_cholesky_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_project_rs_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_bin_tiles": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "gi2d_copy_bins": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "gi2d_rasterize_sum_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp]),
    "gi2d_rasterize_sum_fwd_multi": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _convp, _vp,
                                          _vp]),
    "gi2d_render_cholesky_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_render_rs_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_render_frame": (_i, [ctypes.POINTER(FrameArgs)]),
    "gi2d_render_frame_noalloc": (_i, [ctypes.POINTER(FrameArgs), _i]),
    "gi2d_render_batch": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_fit_backward_step_chain": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp,
                                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          ctypes.POINTER(AdanHyper), _vp]),
    "gi2d_fit_step": (_i, [_vp, _i] + [_vp] * 7 + [_i, _i, _i, _convp] + [_vp] * 11 +
                      [ctypes.POINTER(AdanHyper), _vp]),
    "gi2d_fit_step_batch": (_i, [_vp, _i, _i] + [_vp] * 7 + [_i, _i, _i, _convp] + [_vp] * 11 +
                            [ctypes.POINTER(AdanHyper), _vp]),
    "gi2d_fit_raster": (_i, [_vp, _i] + [_vp] * 7 + [_i, _i, _i, _convp] + [_vp] * 11 + [_vp]),
    "gi2d_fit_tail": (_i, [_vp, _i] + [_vp] * 5 + [_i, _i, _i, _convp] + [_vp] * 12 +
                      [ctypes.POINTER(AdanHyper), _vp]),
    "gi2d_decode_params": (_i, [_i, ctypes.POINTER(Codec), _i, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_decode_render": (_i, [_vp, _i, ctypes.POINTER(Codec), _vp, _vp, _i, _i, _i, _convp, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_fit_forward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gi2d_fit_backward_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AdanHyper), _vp]),
    "gi2d_adan_step": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(AdanHyper), _vp,
                            _vp]),
    "gi2d_rasterize_sum_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _convp, _vp, _vp,
                                    _vp, _vp, _vp, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m gaussianimage_amd.build` "
            "(hipcc, gfx950). This package has no fallback backend.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # The ctypes structures above mirror include/gi2d.h by hand, and the kernels change with
    # their sources: a library built from other sources than the ones next to it is refused.
    from . import build
    if os.path.exists(os.path.join(build.CSRC, build.SOURCES[0])):
        have, want = lib.gi2d_build_id().decode(), build.source_id()
        if have != want and os.environ.get("GI2D_ALLOW_STALE", "0") != "1":
            raise ImportError(
                f"{LIB_PATH} was built from other sources (build id {have}, sources {want}): "
                "rebuild it with `python -m gaussianimage_amd.build` "
                "(GI2D_ALLOW_STALE=1 loads it anyway)")
    return lib


lib = _load()


class Gi2dError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib.gi2d_error_string(status).decode()
        super().__init__(f"{where}: {msg} (status {status})")


def check(status, where):
    if status != GI2D_OK:
        raise Gi2dError(status, where)


def default_conv(**over):
    c = Conv()
    lib.gi2d_default_conv(ctypes.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c
