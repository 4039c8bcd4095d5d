"""Host-side mirror of the reference's operator interface for the hot path.

``project_gaussians_2d``, ``project_gaussians_2d_scale_rot`` and
``rasterize_gaussians_sum`` keep the names, argument order, return tuples and
error behaviour of the operators the reference imports
(/root/reference/gaussianimage_cholesky.py:1-2,77,81-82;
/root/reference/gaussianimage_rs.py:1-2,75-77) and run on the HIP kernels of
``csrc/*.hip`` through the C ABI.  PyTorch is used for device memory, the
current stream and autograd bookkeeping only.

Two bindings of the same three operators exist: the compiled one
(``csrc/torch_ops.cpp`` -> ``_torch_ops.so``: C++ autograd functions, a third of the host
time per call) and the ctypes one in this file.  The compiled binding is used when its
library is present and matches its sources and the running torch; ``use_compiled(False)``
(or ``GI2D_PY_BINDING=1``) selects the ctypes binding.  Both launch the same kernels and
give the same bits; each keeps its own per-stream workspaces.
"""
import ctypes
import os
import threading
import warnings

import torch

from . import _lib
from ._lib import BLOCK, Conv, check, default_conv, lib

__all__ = ["project_gaussians_2d", "project_gaussians_2d_scale_rot", "rasterize_gaussians_sum",
           "rasterize_gaussians_sum_multi", "Workspace", "get_workspace", "set_conventions",
           "get_conventions", "use_compiled", "compiled_binding", "reset_workspace"]

_CONV = default_conv()


def _load_compiled():
    """The compiled binding, or (None, reason) when it is absent or does not match."""
    if os.environ.get("GI2D_PY_BINDING", "0") == "1":
        return None, "GI2D_PY_BINDING=1"
    try:
        from . import _torch_ops as mod
    except ImportError as e:
        return None, "not built (python -m gaussianimage_amd.build): %s" % e
    from . import build
    try:
        want = build.torch_ops_id()
    except OSError:
        want = None  # an installation without the sources
    if want is not None and mod.build_id() != want and \
            os.environ.get("GI2D_ALLOW_STALE", "0") != "1":
        return None, "built from other sources or another torch (%s, now %s); rebuild with " \
                     "python -m gaussianimage_amd.build" % (mod.build_id(), want)
    mod.init(ctypes.addressof(_CONV), _lib.Gi2dError)
    return mod, None


_C, _C_WHY_NOT = _load_compiled()
if _C is None and _C_WHY_NOT != "GI2D_PY_BINDING=1":
    warnings.warn("gaussianimage_amd: the compiled operator binding is not used (%s); the ctypes "
                  "binding runs the same kernels with more host time per call" % _C_WHY_NOT,
                  RuntimeWarning)
_USE_C = _C is not None


def use_compiled(on=True):
    """Selects the compiled (True) or the ctypes (False) binding of the three operators;
    returns the previous choice.  Raises when the compiled binding is not available."""
    global _USE_C
    if on and _C is None:
        raise RuntimeError("the compiled binding is not available: %s" % _C_WHY_NOT)
    prev, _USE_C = _USE_C, bool(on)
    return prev


def compiled_binding():
    """True when the operators currently run through the compiled binding."""
    return _USE_C


def set_conventions(**kw):
    """Override the conventions the reference cannot settle (see ``gi2d_conv``)."""
    for k, v in kw.items():
        if not hasattr(_CONV, k):
            raise ValueError(f"unknown convention {k!r}")
        setattr(_CONV, k, float(v))


def get_conventions():
    return {k: getattr(_CONV, k) for k, _ in Conv._fields_}


def _ptr(t):
    """Device address of a tensor for a ``void*`` parameter (None / empty -> NULL)."""
    return t.data_ptr() if t is not None and t.numel() else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """The current HIP stream of ``device`` as a ``void*`` value."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx) or None
    return torch.cuda.current_stream(device).cuda_stream or None


def _need_cuda(name, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                f"{name}: this operator runs on an AMD GPU through the HIP library only; got a "
                f"{t.device} tensor (there is no CPU fallback)")


def _f32c(t):
    """``t`` as a detached contiguous float32 tensor (itself when it already is one)."""
    if t.requires_grad:
        t = t.detach()
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return t.to(torch.float32).contiguous()


def _late_overflow_text(count, info):
    return ("gaussianimage_amd: an earlier frame overflowed a tile list -- its overflowed tiles "
            "were NaN-filled, so a loss or PSNR computed from that image is NaN (occurrence %d on "
            "this workspace, lists at generation %d, capacity now %d per tile). The lists have "
            "been enlarged; this call is complete. Set GI2D_SYNC=1 (or Workspace.strict = True to "
            "raise instead) to size the lists synchronously in every call."
            % (count, info["generation"], info["capacity"]))


class Workspace:
    """Owns the device-side tile lists of one device (``gi2d_workspace``)."""

    def __init__(self, device):
        self.device = torch.device(device)
        h = ctypes.c_void_p()
        check(lib.gi2d_workspace_create(self.device.index or 0, ctypes.byref(h)),
              "gi2d_workspace_create")
        self.handle = h
        self.key = None       # identity of the tensors the current lists were built from
        self.keep = None      # strong references: keeps their storage from being recycled
        self._counted_key = None
        self._counted_keep = None   # the storage behind counted_key, held for the same reason
        self.rows = (0, 2 ** 31 - 1)
        self.sync_binning = os.environ.get("GI2D_SYNC", "0") == "1"
        self.late_overflows = 0   # GI2D_E_PREV_OVERFLOW reports swallowed by bin()
        self.strict = False       # True: bin() raises on them instead

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                lib.gi2d_workspace_destroy(h)
            except Exception:
                pass

    @property
    def counted_key(self):
        """Identity of the projection outputs whose lists the projection itself prefilled."""
        return self._counted_key

    @counted_key.setter
    def counted_key(self, key):
        self._counted_key = key
        if key is None:
            self._counted_keep = None

    def set_counted(self, xys, radii, H, W):
        """The projection that produced ``xys`` / ``radii`` also filled the lists: remember
        which tensors, and hold their storage (a detached alias, no autograd graph) so that the
        allocator cannot hand the same addresses to other tensors while the key is valid."""
        self._counted_key = self._key(xys, radii, H, W)
        self._counted_keep = (xys.detach(), radii.detach())

    def info(self):
        arr = (ctypes.c_int64 * 10)()
        check(lib.gi2d_workspace_info(self.handle, arr), "gi2d_workspace_info")
        return dict(capacity=arr[0], max_tile_count=arr[1], generation=arr[2], tiles_x=arr[3],
                    tiles_y=arr[4], num_gaussians=arr[5], row_begin=arr[6], row_end=arr[7],
                    frames=arr[8], records=bool(arr[9]))

    def set_rows(self, row_begin=0, row_end=2 ** 31 - 1):
        """Restricts binning/rasterisation to tile rows [row_begin, row_end) (a band)."""
        check(lib.gi2d_workspace_set_rows(self.handle, int(row_begin), int(row_end)),
              "gi2d_workspace_set_rows")
        self.rows = (int(row_begin), int(row_end))
        self.key = None
        self.counted_key = None

    def set_cull(self, on):
        check(lib.gi2d_workspace_set_option(self.handle, _lib.OPT_CULL, int(bool(on))),
              "gi2d_workspace_set_option")

    def set_records(self, on):
        """Per-Gaussian records written by the projections that know colours and opacities
        (default on); off = the rasteriser always gathers the separate tensors (same bits)."""
        check(lib.gi2d_workspace_set_option(self.handle, _lib.OPT_RECORDS, int(bool(on))),
              "gi2d_workspace_set_option")

    def set_raster_ppl(self, ppl):
        """Pixels per lane of the forward rasteriser (1, 2, 4; 0 = automatic)."""
        check(lib.gi2d_workspace_set_option(self.handle, _lib.OPT_RASTER_PPL, int(ppl)),
              "gi2d_workspace_set_option")

    # -- binning ---------------------------------------------------------------------------
    @staticmethod
    def _key(xys, radii, H, W):
        return (xys.data_ptr(), xys._version, radii.data_ptr(), radii._version, xys.shape[0], H, W)

    def bin(self, xys, radii, H, W, flags=0, force=False):
        """Builds (or reuses) the tile lists for these projected Gaussians."""
        key = self._key(xys, radii, H, W)
        if not force and key == self.key:
            return
        tiles_x, tiles_y = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
        if self.sync_binning:
            flags |= _lib.BIN_SYNC
        if self.counted_key == key:
            flags |= _lib.BIN_COUNTED
        self.counted_key = None
        self.key = None
        rc = lib.gi2d_bin_tiles(self.handle, _ptr(xys), _ptr(radii), xys.shape[0], tiles_x,
                                tiles_y, flags, _stream(xys.device))
        if rc == _lib.GI2D_E_PREV_OVERFLOW:
            # An EARLIER asynchronous frame outgrew a tile list: those tiles of ITS image were
            # NaN-filled, the lists have been enlarged since, and the binning asked for now is
            # complete.  The operators the reference calls never fail this way in the middle of
            # a fit, so: count it, say it once, go on (strict=True raises instead).
            self.late_overflows += 1
            if self.strict:
                raise _lib.Gi2dError(rc, "gi2d_bin_tiles")
            # every occurrence is reported (the text differs from one to the next, so the
            # warning filter's once-per-location rule does not hide the later ones)
            warnings.warn(_late_overflow_text(self.late_overflows, self.info()),
                          RuntimeWarning, stacklevel=3)
            rc = _lib.GI2D_OK
        check(rc, "gi2d_bin_tiles")
        self.key = key
        self.keep = (xys, radii)

    def reserve(self, per_tile_capacity, max_tiles=0):
        check(lib.gi2d_workspace_reserve(self.handle, int(per_tile_capacity), int(max_tiles)),
              "gi2d_workspace_reserve")

    def offsets(self):
        """Exclusive scan of the current list lengths, ``tiles + 1`` ints on the device
        (synchronises); the last entry is the number of intersections."""
        inf = self.info()
        T = inf["frames"] * inf["tiles_x"] * (inf["row_end"] - inf["row_begin"])
        offs = torch.empty(T + 1, dtype=torch.int32, device=self.device)
        check(lib.gi2d_copy_bins(self.handle, _ptr(offs), None, 0, _stream(self.device)),
              "gi2d_copy_bins")
        return offs

    def bins(self):
        """(tile_offsets, ids) of the current lists as CPU int32 tensors (synchronises)."""
        st = _stream(self.device)
        offs = self.offsets()
        total = int(offs[-1].item())
        ids = torch.empty(max(total, 1), dtype=torch.int32, device=self.device)
        check(lib.gi2d_copy_bins(self.handle, _ptr(offs), _ptr(ids), ids.numel(), st),
              "gi2d_copy_bins")
        return offs.cpu(), ids[:total].cpu()


class _CompiledWorkspace(Workspace):
    """The same interface over a workspace the compiled binding owns: the list-reuse state
    lives on the C++ side, where the compiled operators read and write it."""

    def __init__(self, device, state):
        self.device = torch.device(device)
        self._st = state
        self.handle = ctypes.c_void_p(state.handle)
        self.rows = (0, 2 ** 31 - 1)

    def __del__(self):
        self.handle = None  # (the compiled binding keeps its workspaces for the process's life)

    key = property(lambda self: self._st.has_lists or None,
                   lambda self, v: self._st.invalidate(False))
    counted_key = property(lambda self: None, lambda self, v: self._st.invalidate(True))
    late_overflows = property(lambda self: self._st.late_overflows,
                              lambda self, v: setattr(self._st, "late_overflows", v))
    strict = property(lambda self: self._st.strict,
                      lambda self, v: setattr(self._st, "strict", bool(v)))
    sync_binning = property(lambda self: self._st.sync_binning,
                            lambda self, v: setattr(self._st, "sync_binning", bool(v)))

    def bin(self, xys, radii, H, W, flags=0, force=False):
        self._st.bin(xys, radii, int(H), int(W), int(flags), bool(force))


_WORKSPACES = {}
_WS_LOCK = threading.Lock()


def get_workspace(device):
    """The operators' workspace for the CURRENT STREAM of ``device``.  A workspace serves one
    stream at a time (its lists are filled by one launch and read by the next), so every stream
    that calls the operators gets its own: two streams -- two models, an evaluation beside a fit
    -- never share tile lists.  (Torch hands out a bounded pool of streams per device.)"""
    if not isinstance(device, torch.device):
        device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = _stream(device) or 0
    key = (idx, stream, _USE_C)  # (each binding has its own)
    ws = _WORKSPACES.get(key)  # (entries are only ever added)
    if ws is not None:
        return ws
    with _WS_LOCK:
        ws = _WORKSPACES.get(key)
        if ws is None:
            dev = torch.device("cuda", idx)
            ws = _WORKSPACES[key] = (_CompiledWorkspace(dev, _C.workspace(idx, stream)) if _USE_C
                                     else Workspace(dev))
        return ws


def reset_workspace(device):
    """Drops the operators' workspace of the current stream of ``device`` (both bindings): the
    next call starts from a fresh one, with lists sized anew.  The caller makes sure no work
    that uses the old lists is still queued."""
    if not isinstance(device, torch.device):
        device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = _stream(device) or 0
    with _WS_LOCK:
        _WORKSPACES.pop((idx, stream, False), None)
        _WORKSPACES.pop((idx, stream, True), None)
    if _C is not None:
        _C.drop_workspace(idx, stream)


def _tile_bounds_check(name, H, W, tile_bounds):
    tx, ty = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    if int(tile_bounds[0]) != tx or int(tile_bounds[1]) != ty:
        raise ValueError(f"{name}: tile_bounds {tuple(tile_bounds)} does not match a {W}x{H} image "
                         f"with {BLOCK}x{BLOCK} tiles (expected ({tx}, {ty}, 1))")
    return tx, ty


class _NoGradCtx:
    """Stands in for the autograd context when nothing is to be recorded."""
    hw = opacity_shape = None

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


_NO_GRAD_CTX = _NoGradCtx()


def _call(fn, differentiable, *args):
    """``fn.apply(*args)``; under ``torch.no_grad()`` (the reference's FPS and test loops,
    train.py:242,258) or when no differentiable input requires a gradient, the forward runs
    directly: the autograd bookkeeping of ``apply`` is a third of such a call's host time."""
    if torch.is_grad_enabled():
        for t in differentiable:
            if t.requires_grad:
                return fn.apply(*args)
    return fn.forward(_NO_GRAD_CTX, *args)


def _alloc_proj(N, device):
    return (torch.empty(N, 2, dtype=torch.float32, device=device),
            torch.empty(N, dtype=torch.float32, device=device),
            torch.empty(N, dtype=torch.int32, device=device),
            torch.empty(N, 3, dtype=torch.float32, device=device),
            torch.empty(N, dtype=torch.int32, device=device))


class _ProjectGaussians2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, L_elements, img_height, img_width, tile_bounds):
        _need_cuda("project_gaussians_2d", means2d, L_elements)
        tx, ty = _tile_bounds_check("project_gaussians_2d", img_height, img_width, tile_bounds)
        m, L = _f32c(means2d), _f32c(L_elements)
        N = m.shape[0]
        xys, depths, radii, conics, nth = _alloc_proj(N, m.device)
        ws = get_workspace(m.device)
        ws.key = None
        check(lib.gi2d_project_cholesky_fwd(_ptr(m), _ptr(L), N, img_height, img_width, tx, ty,
                                            ctypes.byref(_CONV), _ptr(xys), _ptr(depths),
                                            _ptr(radii), _ptr(conics), _ptr(nth), ws.handle,
                                            _stream(m.device)), "gi2d_project_cholesky_fwd")
        ws.set_counted(xys, radii, img_height, img_width)
        ctx.save_for_backward(m, L)
        ctx.hw = (img_height, img_width)
        ctx.mark_non_differentiable(depths, radii, nth)
        return xys, depths, radii, conics, nth

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_nth):
        m, L = ctx.saved_tensors
        H, W = ctx.hw
        N = m.shape[0]
        v_xys = _f32c(v_xys) if v_xys is not None else torch.zeros_like(m)
        v_conics = _f32c(v_conics) if v_conics is not None else torch.zeros_like(L)
        v_m, v_L = torch.empty_like(m), torch.empty_like(L)
        check(lib.gi2d_project_cholesky_bwd(_ptr(m), _ptr(L), N, H, W, _ptr(v_xys),
                                            _ptr(v_conics), _ptr(v_m), _ptr(v_L),
                                            _stream(m.device)), "gi2d_project_cholesky_bwd")
        return v_m, v_L, None, None, None


class _ProjectGaussians2dScaleRot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, scales2d, rotation, img_height, img_width, tile_bounds):
        _need_cuda("project_gaussians_2d_scale_rot", means2d, scales2d, rotation)
        tx, ty = _tile_bounds_check("project_gaussians_2d_scale_rot", img_height, img_width,
                                    tile_bounds)
        m, s, r = _f32c(means2d), _f32c(scales2d), _f32c(rotation)
        N = m.shape[0]
        xys, depths, radii, conics, nth = _alloc_proj(N, m.device)
        ws = get_workspace(m.device)
        ws.key = None
        check(lib.gi2d_project_rs_fwd(_ptr(m), _ptr(s), _ptr(r), N, img_height, img_width, tx, ty,
                                      ctypes.byref(_CONV), _ptr(xys), _ptr(depths), _ptr(radii),
                                      _ptr(conics), _ptr(nth), ws.handle, _stream(m.device)),
              "gi2d_project_rs_fwd")
        ws.set_counted(xys, radii, img_height, img_width)
        ctx.save_for_backward(m, s, r)
        ctx.hw = (img_height, img_width)
        ctx.mark_non_differentiable(depths, radii, nth)
        return xys, depths, radii, conics, nth

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_nth):
        m, s, r = ctx.saved_tensors
        H, W = ctx.hw
        N = m.shape[0]
        v_xys = _f32c(v_xys) if v_xys is not None else torch.zeros_like(m)
        v_conics = (_f32c(v_conics) if v_conics is not None else
                    torch.zeros(N, 3, dtype=torch.float32, device=m.device))
        v_m, v_s, v_r = torch.empty_like(m), torch.empty_like(s), torch.empty_like(r)
        check(lib.gi2d_project_rs_bwd(_ptr(m), _ptr(s), _ptr(r), N, H, W, _ptr(v_xys),
                                      _ptr(v_conics), _ptr(v_m), _ptr(v_s), _ptr(v_r),
                                      _stream(m.device)), "gi2d_project_rs_bwd")
        return v_m, v_s, v_r, None, None, None


class _RasterizeGaussiansSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                img_width, background):
        dev = xys.device
        N = xys.shape[0]
        # the tile lists are keyed on the caller's own xys/radii tensors, so that the passes of
        # one model forward (gaussianimage_cholesky.py:81,89,95) share one binning
        xys_c = xys.detach()
        if xys_c.dtype != torch.float32 or not xys_c.is_contiguous():
            xys_c = _f32c(xys)
        radii_c = radii if (radii.dtype == torch.int32 and radii.is_contiguous()) else \
            radii.to(torch.int32).contiguous()
        con, col = _f32c(conics), _f32c(colors)
        opa = _f32c(opacity).reshape(-1)
        bg = _f32c(background)
        ws = get_workspace(dev)
        ws.bin(xys_c, radii_c, img_height, img_width)
        out = torch.empty(img_height, img_width, 3, dtype=torch.float32, device=dev)
        check(lib.gi2d_rasterize_sum_fwd(ws.handle, _ptr(xys_c), _ptr(con), _ptr(col), _ptr(opa),
                                         _ptr(bg), N, img_height, img_width, ctypes.byref(_CONV),
                                         _ptr(out), _stream(dev)), "gi2d_rasterize_sum_fwd")
        ctx.save_for_backward(xys_c, radii_c, con, col, opa)
        ctx.hw = (img_height, img_width)
        ctx.opacity_shape = opacity.shape
        return out

    @staticmethod
    def backward(ctx, v_out):
        xys, radii, con, col, opa = ctx.saved_tensors
        H, W = ctx.hw
        N = xys.shape[0]
        dev = xys.device
        v_out = _f32c(v_out)
        ws = get_workspace(dev)
        ws.bin(xys, radii, H, W)  # reuses the forward's lists unless another frame replaced them
        v_xy = torch.empty(N, 2, dtype=torch.float32, device=dev)
        v_con = torch.empty(N, 3, dtype=torch.float32, device=dev)
        v_col = torch.empty(N, 3, dtype=torch.float32, device=dev)
        v_opa = torch.empty(N, dtype=torch.float32, device=dev)
        check(lib.gi2d_rasterize_sum_bwd(ws.handle, _ptr(xys), _ptr(radii), _ptr(con), _ptr(col),
                                         _ptr(opa), N, H, W, ctypes.byref(_CONV), _ptr(v_out),
                                         _ptr(v_xy), _ptr(v_con), _ptr(v_col), _ptr(v_opa),
                                         _stream(dev)), "gi2d_rasterize_sum_bwd")
        return (v_xy, None, None, v_con, None, v_col, v_opa.reshape(ctx.opacity_shape), None, None,
                None)


class _RasterizeGaussiansSumMulti(torch.autograd.Function):
    """Several colour sets over one set of Gaussians in one forward launch; the backward runs
    the single-set backward once per set that received a gradient and adds the geometry terms."""

    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width,
                background, *colors):
        dev = xys.device
        N = xys.shape[0]
        xys_c = xys.detach()
        if xys_c.dtype != torch.float32 or not xys_c.is_contiguous():
            xys_c = _f32c(xys)
        radii_c = radii if (radii.dtype == torch.int32 and radii.is_contiguous()) else \
            radii.to(torch.int32).contiguous()
        con = _f32c(conics)
        cols = [_f32c(c) for c in colors]
        opa = _f32c(opacity).reshape(-1)
        bg = _f32c(background)
        ws = get_workspace(dev)
        ws.bin(xys_c, radii_c, img_height, img_width)
        outs = [torch.empty(img_height, img_width, 3, dtype=torch.float32, device=dev)
                for _ in cols]
        S = len(cols)
        col_ptrs = (ctypes.c_void_p * S)(*[c.data_ptr() for c in cols])
        out_ptrs = (ctypes.c_void_p * S)(*[o.data_ptr() for o in outs])
        check(lib.gi2d_rasterize_sum_fwd_multi(ws.handle, _ptr(xys_c), _ptr(con), S, col_ptrs,
                                               _ptr(opa), _ptr(bg), N, img_height, img_width,
                                               ctypes.byref(_CONV), out_ptrs, _stream(dev)),
              "gi2d_rasterize_sum_fwd_multi")
        ctx.save_for_backward(xys_c, radii_c, con, opa, *cols)
        ctx.hw = (img_height, img_width)
        ctx.opacity_shape = opacity.shape
        return tuple(outs)

    @staticmethod
    def backward(ctx, *v_outs):
        xys, radii, con, opa, *cols = ctx.saved_tensors
        H, W = ctx.hw
        N = xys.shape[0]
        dev = xys.device
        ws = get_workspace(dev)
        ws.bin(xys, radii, H, W)
        f32 = torch.float32
        v_xy = v_con = v_opa = None
        v_cols = []
        for col, v_out in zip(cols, v_outs):
            if v_out is None:
                v_cols.append(None)
                continue
            v_out = _f32c(v_out)
            g_xy = torch.empty(N, 2, dtype=f32, device=dev)
            g_con = torch.empty(N, 3, dtype=f32, device=dev)
            g_col = torch.empty(N, 3, dtype=f32, device=dev)
            g_opa = torch.empty(N, dtype=f32, device=dev)
            check(lib.gi2d_rasterize_sum_bwd(ws.handle, _ptr(xys), _ptr(radii), _ptr(con),
                                             _ptr(col), _ptr(opa), N, H, W, ctypes.byref(_CONV),
                                             _ptr(v_out), _ptr(g_xy), _ptr(g_con), _ptr(g_col),
                                             _ptr(g_opa), _stream(dev)), "gi2d_rasterize_sum_bwd")
            v_cols.append(g_col)
            if v_xy is None:
                v_xy, v_con, v_opa = g_xy, g_con, g_opa
            else:
                v_xy, v_con, v_opa = v_xy + g_xy, v_con + g_con, v_opa + g_opa
        if v_opa is not None:
            v_opa = v_opa.reshape(ctx.opacity_shape)
        return (v_xy, None, None, v_con, None, v_opa, None, None, None, *v_cols)


def project_gaussians_2d(means2d, L_elements, img_height, img_width, tile_bounds):
    """``gsplat.project_gaussians_2d.project_gaussians_2d`` of the reference
    (/root/reference/gaussianimage_cholesky.py:77).  Returns
    ``(xys[N,2], depths[N], radii[N] int32, conics[N,3], num_tiles_hit[N] int32)``."""
    if _USE_C:
        return _C.project_gaussians_2d(means2d, L_elements, img_height, img_width, tile_bounds)
    if means2d.ndimension() != 2 or means2d.shape[1] != 2:
        raise ValueError("means2d must have shape (N, 2)")
    if L_elements.ndimension() != 2 or L_elements.shape != (means2d.shape[0], 3):
        raise ValueError("L_elements must have shape (N, 3)")
    return _call(_ProjectGaussians2d, (means2d, L_elements), means2d, L_elements,
                 int(img_height), int(img_width), tuple(tile_bounds))


def project_gaussians_2d_scale_rot(means2d, scales2d, rotation, img_height, img_width,
                                   tile_bounds):
    """``gsplat.project_gaussians_2d_scale_rot.project_gaussians_2d_scale_rot``
    (/root/reference/gaussianimage_rs.py:75); ``rotation`` in radians, shape (N, 1)."""
    if _USE_C:
        return _C.project_gaussians_2d_scale_rot(means2d, scales2d, rotation, img_height,
                                                 img_width, tile_bounds)
    N = means2d.shape[0]
    if means2d.ndimension() != 2 or means2d.shape[1] != 2:
        raise ValueError("means2d must have shape (N, 2)")
    if scales2d.ndimension() != 2 or scales2d.shape != (N, 2):
        raise ValueError("scales2d must have shape (N, 2)")
    if rotation.numel() != N:
        raise ValueError("rotation must have shape (N, 1)")
    return _call(_ProjectGaussians2dScaleRot, (means2d, scales2d, rotation), means2d, scales2d,
                 rotation, int(img_height), int(img_width), tuple(tile_bounds))


def rasterize_gaussians_sum(xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                            img_height, img_width, BLOCK_H=16, BLOCK_W=16, background=None,
                            return_alpha=False):
    """``gsplat.rasterize_sum.rasterize_gaussians_sum`` of the reference
    (/root/reference/gaussianimage_cholesky.py:81-82): the accumulated sum
    ``out[p] = sum_g colors[g] * min(cap, opacity[g] * exp(-sigma_g(p)))`` over the Gaussians
    binned to the pixel's tile; ``[H, W, D]`` float32.  The kernels render three channels per
    colour set and up to three sets per launch: D = 3 (every call site of the reference) is one
    launch; other D run ceil(D / 9) launches over zero-padded sets.  ``return_alpha=True`` adds
    ``alpha[H, W]``, the accumulated weight ``sum_g min(cap, opacity[g] * exp(-sigma_g(p)))``
    (the alpha of a sum has no transmittance meaning), from the same launch as the image."""
    if colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")
    if opacity.ndimension() != 2:
        raise ValueError("opacity must have dimensions (N, 1)")
    if return_alpha or colors.shape[1] != 3:
        return _rasterize_general(xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                  img_height, img_width, BLOCK_H, BLOCK_W, background, return_alpha)
    if _USE_C:
        return _C.rasterize_gaussians_sum(xys, radii, conics, colors, opacity, img_height,
                                          img_width, BLOCK_H, BLOCK_W, background)
    if int(BLOCK_H) != BLOCK or int(BLOCK_W) != BLOCK:
        raise ValueError(f"tile size is fixed at {BLOCK}x{BLOCK} (the reference's train.py:54)")
    _need_cuda("rasterize_gaussians_sum", xys, conics, colors, opacity)
    if background is None:
        background = torch.ones(3, dtype=torch.float32, device=xys.device)
    if background.shape[0] != colors.shape[1]:
        raise ValueError("background and colors must have the same number of channels")
    N = xys.shape[0]
    for name, t in (("radii", radii), ("conics", conics), ("colors", colors),
                    ("opacity", opacity)):
        if t.shape[0] != N:
            raise ValueError(f"{name} has {t.shape[0]} rows, xys has {N}")
    return _call(_RasterizeGaussiansSum, (xys, conics, colors, opacity), xys, depths, radii,
                 conics, num_tiles_hit, colors, opacity, int(img_height), int(img_width),
                 background)


def _rasterize_general(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                       img_width, BLOCK_H, BLOCK_W, background, return_alpha):
    """Any number of channels, and the accumulated weight, through the multi-set rasteriser."""
    N, D = colors.shape
    if D < 1:
        raise ValueError("colors must have at least one channel")
    if background is None:
        background = torch.ones(D, dtype=torch.float32, device=xys.device)
    if background.shape[0] != D:
        raise ValueError("background and colors must have the same number of channels")
    padded = torch.nn.functional.pad(colors, (0, (-D) % 3)) if D % 3 else colors
    sets = [padded[:, k:k + 3] for k in range(0, padded.shape[1], 3)]
    if return_alpha:
        sets.append(torch.ones(N, 3, dtype=torch.float32, device=colors.device))
    # (the kernels add ONE 3-channel background to every set of a launch: it is added here)
    no_bg = torch.zeros(3, dtype=torch.float32, device=xys.device)
    outs = []
    for k in range(0, len(sets), _lib.MAX_COLOR_SETS):
        outs += list(rasterize_gaussians_sum_multi(
            xys, depths, radii, conics, num_tiles_hit, sets[k:k + _lib.MAX_COLOR_SETS], opacity,
            img_height, img_width, BLOCK_H, BLOCK_W, background=no_bg))
    alpha = outs.pop()[..., 0] if return_alpha else None
    img = outs[0] if D == 3 else torch.cat(outs, dim=-1)[..., :D]
    if _CONV.bg_weight != 0.0:
        img = img + _CONV.bg_weight * background.to(img.dtype)
    return (img, alpha) if return_alpha else img


def rasterize_gaussians_sum_multi(xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                  img_height, img_width, BLOCK_H=16, BLOCK_W=16, background=None):
    """``rasterize_gaussians_sum`` for a list of up to 3 colour tensors ``[N,3]`` over the same
    Gaussians, in one launch; returns a tuple of ``[H,W,3]`` images, each bit-identical to the
    single-set call.  No counterpart in the reference's ``gsplat``: it is what the three
    back-to-back passes of ``GaussianImage_Cholesky.forward``
    (/root/reference/gaussianimage_cholesky.py:81-97) amount to.  Differentiable."""
    colors = list(colors)
    if not 1 <= len(colors) <= _lib.MAX_COLOR_SETS:
        raise ValueError("between 1 and %d colour sets" % _lib.MAX_COLOR_SETS)
    if opacity.ndimension() != 2:
        raise ValueError("opacity must have dimensions (N, 1)")
    N = xys.shape[0]
    for c in colors:
        if c.ndimension() != 2 or c.shape[1] != 3 or c.shape[0] != N:
            raise ValueError("every colour set must have shape (N, 3)")
    if int(BLOCK_H) != BLOCK or int(BLOCK_W) != BLOCK:
        raise ValueError(f"tile size is fixed at {BLOCK}x{BLOCK} (the reference's train.py:54)")
    _need_cuda("rasterize_gaussians_sum_multi", xys, conics, opacity, *colors)
    if background is None:
        background = torch.ones(3, dtype=torch.float32, device=xys.device)
    if background.shape[0] != 3:
        raise ValueError("background and colors must have the same number of channels")
    for name, t in (("radii", radii), ("conics", conics), ("opacity", opacity)):
        if t.shape[0] != N:
            raise ValueError(f"{name} has {t.shape[0]} rows, xys has {N}")
    return _call(_RasterizeGaussiansSumMulti, (xys, conics, opacity, *colors), xys, depths,
                 radii, conics, num_tiles_hit, opacity, int(img_height), int(img_width),
                 background, *colors)
