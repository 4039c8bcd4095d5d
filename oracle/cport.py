"""ctypes front end of ``gi2d_oracle.c`` (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``; never by the product package.
PARITY UNPINNED -- see the header of ``gi2d_oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgi2d_oracle.so")
BLOCK = 16


class Conv(ctypes.Structure):
    _fields_ = [("pixel_offset", ctypes.c_float), ("center_bias", ctypes.c_float),
                ("alpha_cap", ctypes.c_float), ("alpha_min", ctypes.c_float),
                ("bg_weight", ctypes.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_bin_tiles.restype = ctypes.c_int64
    return _lib


def default_conv(**over):
    c = Conv()
    lib().oracle_default_conv(ctypes.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def tile_bounds(H, W):
    return ((W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK, 1)


def _alloc_proj(N):
    return (np.empty((N, 2), np.float32), np.empty(N, np.float32), np.empty(N, np.int32),
            np.empty((N, 3), np.float32), np.empty(N, np.int32))


def project_cholesky_fwd(means, L, H, W, conv=None):
    conv = conv or default_conv()
    means, L = _f(means), _f(L)
    N = means.shape[0]
    tb = tile_bounds(H, W)
    xys, depths, radii, conics, nth = _alloc_proj(N)
    lib().oracle_project_cholesky_fwd(N, _p(means), _p(L), H, W, tb[0], tb[1], ctypes.byref(conv),
                                      _p(xys), _p(depths), _p(radii), _p(conics), _p(nth))
    return xys, depths, radii, conics, nth


def project_rs_fwd(means, scales, rot, H, W, conv=None):
    conv = conv or default_conv()
    means, scales, rot = _f(means), _f(scales), _f(rot)
    N = means.shape[0]
    tb = tile_bounds(H, W)
    xys, depths, radii, conics, nth = _alloc_proj(N)
    lib().oracle_project_rs_fwd(N, _p(means), _p(scales), _p(rot), H, W, tb[0], tb[1],
                                ctypes.byref(conv), _p(xys), _p(depths), _p(radii), _p(conics),
                                _p(nth))
    return xys, depths, radii, conics, nth


def bin_tiles(xys, radii, H, W):
    xys, radii = _f(xys), _i(radii)
    N = xys.shape[0]
    tb = tile_bounds(H, W)
    T = tb[0] * tb[1]
    offs = np.empty(T + 1, np.int32)
    total = lib().oracle_bin_tiles(N, _p(xys), _p(radii), tb[0], tb[1], _p(offs), None,
                                   ctypes.c_int64(0))
    ids = np.empty(max(total, 1), np.int32)
    lib().oracle_bin_tiles(N, _p(xys), _p(radii), tb[0], tb[1], _p(offs), _p(ids),
                           ctypes.c_int64(ids.size))
    return offs, ids[:total]


def rasterize_sum_fwd(xys, conics, colors, opacity, H, W, offs, ids, background=None, conv=None,
                      want_margin=False):
    conv = conv or default_conv()
    xys, conics, colors, opacity = _f(xys), _f(conics), _f(colors), _f(opacity).reshape(-1)
    bg = _f(background if background is not None else np.zeros(3))
    tb = tile_bounds(H, W)
    out = np.empty((H, W, 3), np.float32)
    margin = np.empty((H, W), np.float32) if want_margin else None
    ids = _i(ids) if len(ids) else np.zeros(1, np.int32)
    lib().oracle_rasterize_sum_fwd(xys.shape[0], _p(xys), _p(conics), _p(colors), _p(opacity),
                                   _p(bg), H, W, tb[0], tb[1], _p(_i(offs)), _p(ids),
                                   ctypes.byref(conv), _p(out),
                                   _p(margin) if want_margin else None)
    return (out, margin) if want_margin else out


def rasterize_sum_bwd(xys, conics, colors, opacity, H, W, offs, ids, v_out, conv=None):
    conv = conv or default_conv()
    xys, conics, colors, opacity = _f(xys), _f(conics), _f(colors), _f(opacity).reshape(-1)
    v_out = _f(v_out)
    N = xys.shape[0]
    tb = tile_bounds(H, W)
    v_xy = np.empty((N, 2), np.float32)
    v_conic = np.empty((N, 3), np.float32)
    v_colors = np.empty((N, 3), np.float32)
    v_opacity = np.empty((N, 1), np.float32)
    ids = _i(ids) if len(ids) else np.zeros(1, np.int32)
    lib().oracle_rasterize_sum_bwd(N, _p(xys), _p(conics), _p(colors), _p(opacity), H, W, tb[0],
                                   tb[1], _p(_i(offs)), _p(ids), ctypes.byref(conv), _p(v_out),
                                   _p(v_xy), _p(v_conic), _p(v_colors), _p(v_opacity))
    return v_xy, v_conic, v_colors, v_opacity


def project_cholesky_bwd(means, L, H, W, v_xy, v_conic):
    means, L, v_xy, v_conic = _f(means), _f(L), _f(v_xy), _f(v_conic)
    N = means.shape[0]
    v_means = np.empty((N, 2), np.float32)
    v_L = np.empty((N, 3), np.float32)
    lib().oracle_project_cholesky_bwd(N, _p(means), _p(L), H, W, _p(v_xy), _p(v_conic),
                                      _p(v_means), _p(v_L))
    return v_means, v_L


def project_rs_bwd(means, scales, rot, H, W, v_xy, v_conic):
    means, scales, rot = _f(means), _f(scales), _f(rot)
    v_xy, v_conic = _f(v_xy), _f(v_conic)
    N = means.shape[0]
    v_means = np.empty((N, 2), np.float32)
    v_scales = np.empty((N, 2), np.float32)
    v_rot = np.empty((N, 1), np.float32)
    lib().oracle_project_rs_bwd(N, _p(means), _p(scales), _p(rot), H, W, _p(v_xy), _p(v_conic),
                                _p(v_means), _p(v_scales), _p(v_rot))
    return v_means, v_scales, v_rot


def render_cholesky(means, L, colors, H, W, opacity=None, conv=None):
    """project -> bin -> rasterize; the forward hot path end to end."""
    xys, _, radii, conics, _ = project_cholesky_fwd(means, L, H, W, conv)
    offs, ids = bin_tiles(xys, radii, H, W)
    if opacity is None:
        opacity = np.ones((xys.shape[0], 1), np.float32)
    return rasterize_sum_fwd(xys, conics, colors, opacity, H, W, offs, ids, conv=conv)


def sincos_f32(theta):
    """The shared fixed-sequence sine / cosine (csrc/sincos_f32.h) on a float32 array."""
    theta = _f(theta).reshape(-1)
    s, c = np.empty_like(theta), np.empty_like(theta)
    lib().oracle_sincos_f32(theta.size, _p(theta), _p(s), _p(c))
    return s, c
