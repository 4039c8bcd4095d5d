"""Allocation-free driver of the hot path: project -> bin -> rasterize_sum (and back).

``FramePipeline`` owns every buffer of one frame configuration (operator inputs are handed in
as device tensors) and issues the C-ABI calls back to back on the current stream without host
synchronisation.  It is what ``bench.py`` times and what the tile-row sharding builds on; the
autograd operators in ``ops.py`` are the drop-in boundary for the reference's models.
"""
import ctypes

import torch

from . import _lib
from ._lib import BLOCK, check, lib
from .ops import Workspace, _ptr, _stream, get_conventions

__all__ = ["FramePipeline", "FrameRing", "FrameBatch"]


class FramePipeline:
    """One frame configuration: ``N`` Gaussians on an ``H x W`` image, Cholesky or RS.

    ``rows=(r0, r1)`` restricts binning and rasterisation to tile rows ``[r0, r1)`` of the
    frame (tile-row sharding); the image buffer then holds only that band.
    """

    def __init__(self, num_points, H, W, kind="cholesky", device="cuda:0", rows=None,
                 workspace=None):
        if kind not in ("cholesky", "rs"):
            raise ValueError("kind must be 'cholesky' or 'rs'")
        self.N, self.H, self.W, self.kind = int(num_points), int(H), int(W), kind
        self.device = torch.device(device)
        self.tiles_x = (self.W + BLOCK - 1) // BLOCK
        self.tiles_y = (self.H + BLOCK - 1) // BLOCK
        self.rows = (0, self.tiles_y) if rows is None else (int(rows[0]), int(rows[1]))
        if not (0 <= self.rows[0] <= self.rows[1] <= self.tiles_y):
            raise ValueError(f"rows {self.rows} outside [0, {self.tiles_y}]")
        self.band_y0 = self.rows[0] * BLOCK
        self.band_h = max(0, min(self.H, self.rows[1] * BLOCK) - self.band_y0)
        self.ws = workspace or Workspace(self.device)
        self.conv = _lib.default_conv(**get_conventions())
        d, N = self.device, self.N
        f32, i32 = torch.float32, torch.int32
        self.xys = torch.empty(N, 2, dtype=f32, device=d)
        self.depths = torch.empty(N, dtype=f32, device=d)
        self.radii = torch.empty(N, dtype=i32, device=d)
        self.conics = torch.empty(N, 3, dtype=f32, device=d)
        self.num_tiles_hit = torch.empty(N, dtype=i32, device=d)
        self.image = torch.empty(self.band_h, self.W, 3, dtype=f32, device=d)
        self.background = torch.ones(3, dtype=f32, device=d)
        # backward buffers are made on first use
        self.v_xys = self.v_conics = self.v_colors = self.v_opacity = None
        self.v_means = self.v_cov_a = self.v_cov_b = None
        self._set_rows()

    def _set_rows(self):
        self.ws.set_rows(self.rows[0], self.rows[1])

    # -- forward ---------------------------------------------------------------------------
    def project(self, means, cov_a, cov_b=None):
        """``cov_a`` = L_elements[N,3] (Cholesky) or scales[N,2] with ``cov_b`` = rotation[N,1]."""
        st = _stream(self.device)
        outs = (_ptr(self.xys), _ptr(self.depths), _ptr(self.radii), _ptr(self.conics),
                _ptr(self.num_tiles_hit))
        if self.kind == "cholesky":
            check(lib.gi2d_project_cholesky_fwd(_ptr(means), _ptr(cov_a), self.N, self.H, self.W,
                                                self.tiles_x, self.tiles_y,
                                                ctypes.byref(self.conv), *outs, self.ws.handle,
                                                st), "gi2d_project_cholesky_fwd")
        else:
            check(lib.gi2d_project_rs_fwd(_ptr(means), _ptr(cov_a), _ptr(cov_b), self.N, self.H,
                                          self.W, self.tiles_x, self.tiles_y,
                                          ctypes.byref(self.conv), *outs, self.ws.handle, st),
                  "gi2d_project_rs_fwd")

    def bin(self, flags=_lib.BIN_COUNTED):
        check(lib.gi2d_bin_tiles(self.ws.handle, _ptr(self.xys), _ptr(self.radii), self.N,
                                 self.tiles_x, self.tiles_y, flags, _stream(self.device)),
              "gi2d_bin_tiles")
        self.ws.key = None

    def rasterize(self, colors, opacity, out=None):
        out = self.image if out is None else out
        check(lib.gi2d_rasterize_sum_fwd(self.ws.handle, _ptr(self.xys), _ptr(self.conics),
                                         _ptr(colors), _ptr(opacity), _ptr(self.background),
                                         self.N, self.H, self.W, ctypes.byref(self.conv),
                                         _ptr(out), _stream(self.device)),
              "gi2d_rasterize_sum_fwd")
        return out

    def rasterize_multi(self, colors, opacity, outs=None):
        """Up to 3 colour tensors ``[N,3]`` over the current lists in one launch; returns the
        list of ``[band_h, W, 3]`` images (the first defaults to the pipeline's own buffer)."""
        S = len(colors)
        if outs is None:
            if len(getattr(self, "extra_images", ())) < S - 1:
                self.extra_images = [torch.empty_like(self.image) for _ in range(S - 1)]
            outs = [self.image] + self.extra_images[:S - 1]
        col_ptrs = (ctypes.c_void_p * S)(*[c.data_ptr() for c in colors])
        out_ptrs = (ctypes.c_void_p * S)(*[o.data_ptr() for o in outs])
        check(lib.gi2d_rasterize_sum_fwd_multi(self.ws.handle, _ptr(self.xys), _ptr(self.conics),
                                               S, col_ptrs, _ptr(opacity), _ptr(self.background),
                                               self.N, self.H, self.W, ctypes.byref(self.conv),
                                               out_ptrs, _stream(self.device)),
              "gi2d_rasterize_sum_fwd_multi")
        return outs

    def forward(self, means, cov_a, cov_b, colors, opacity, out=None, events=None):
        """project -> bin -> rasterize in one C call; returns the (band of the) image,
        ``[band_h, W, 3]``.  ``events``: optional pair of ``torch.cuda.Event`` recorded right
        before / after the rasterize launch."""
        out = self.image if out is None else out
        if self.rows_dirty():
            self._set_rows()
        ev0 = ev1 = None
        if events is not None:
            for e in events:  # a torch event has no handle until it is first recorded
                if not e.cuda_event:
                    e.record()
            ev0, ev1 = (ctypes.c_void_p(e.cuda_event) for e in events)
        common = (_ptr(colors), _ptr(opacity), _ptr(self.background), self.N, self.H, self.W,
                  ctypes.byref(self.conv), _ptr(self.xys), _ptr(self.depths), _ptr(self.radii),
                  _ptr(self.conics), _ptr(self.num_tiles_hit), _ptr(out), ev0, ev1,
                  _stream(self.device))
        if self.kind == "cholesky":
            rc = lib.gi2d_render_cholesky_fwd(self.ws.handle, _ptr(means), _ptr(cov_a), *common)
        else:
            rc = lib.gi2d_render_rs_fwd(self.ws.handle, _ptr(means), _ptr(cov_a), _ptr(cov_b),
                                        *common)
        check(rc, "gi2d_render_%s_fwd" % self.kind)
        self.ws.key = None
        return out

    def bind(self, means, cov_a, cov_b, colors, opacity, out=None):
        """Pre-builds the argument list of the per-frame call for a fixed set of tensors and
        returns ``run(events=None)``: the same as ``forward`` without the per-call Python work
        (for render loops over persistent buffers).  The tensors must stay alive and in place."""
        out = self.image if out is None else out
        args = _lib.FrameArgs(
            ws=self.ws.handle, kind=0 if self.kind == "cholesky" else 1, means2d=_ptr(means),
            cov_a=_ptr(cov_a), cov_b=_ptr(cov_b) if self.kind != "cholesky" else None,
            colors=_ptr(colors), opacity=_ptr(opacity), background=_ptr(self.background),
            N=self.N, H=self.H, W=self.W, conv=self.conv, xys=_ptr(self.xys),
            depths=_ptr(self.depths), radii=_ptr(self.radii), conics=_ptr(self.conics),
            num_tiles_hit=_ptr(self.num_tiles_hit), out_img=_ptr(out), raster_begin_event=None,
            raster_end_event=None, stream=_stream(self.device))
        keep = (means, cov_a, cov_b, colors, opacity, out, args)
        ref = ctypes.byref(args)
        fn = lib.gi2d_render_frame
        ws, rows, name = self.ws, self.rows, "gi2d_render_%s_fwd" % self.kind

        def run(events=None, _keep=keep):
            if ws.rows != rows:
                ws.set_rows(*rows)
            if events is None:
                rc = fn(ref)
            else:
                args.raster_begin_event = events[0].cuda_event
                args.raster_end_event = events[1].cuda_event
                rc = fn(ref)
                args.raster_begin_event = args.raster_end_event = None
            if rc:
                check(rc, name)
            ws.key = None
            return out

        return run

    def rows_dirty(self):
        """True when another user of the (shared) workspace may have changed its tile rows."""
        return self.ws.rows != self.rows

    # -- backward --------------------------------------------------------------------------
    def _alloc_backward(self, cov_a, cov_b):
        d, N, f32 = self.device, self.N, torch.float32
        self.v_xys = torch.empty(N, 2, dtype=f32, device=d)
        self.v_conics = torch.empty(N, 3, dtype=f32, device=d)
        self.v_colors = torch.empty(N, 3, dtype=f32, device=d)
        self.v_opacity = torch.empty(N, dtype=f32, device=d)
        self.v_means = torch.empty(N, 2, dtype=f32, device=d)
        self.v_cov_a = torch.empty_like(cov_a)
        self.v_cov_b = torch.empty_like(cov_b) if cov_b is not None else None

    def backward(self, means, cov_a, cov_b, colors, opacity, v_image):
        """Gradients of ``sum(image * v_image)``: returns
        ``(v_means, v_cov_a, v_cov_b, v_colors, v_opacity)`` (buffers owned by the pipeline)."""
        if self.v_xys is None:
            self._alloc_backward(cov_a, cov_b)
        st = _stream(self.device)
        check(lib.gi2d_rasterize_sum_bwd(self.ws.handle, _ptr(self.xys), _ptr(self.radii),
                                         _ptr(self.conics), _ptr(colors), _ptr(opacity), self.N,
                                         self.H, self.W, ctypes.byref(self.conv), _ptr(v_image),
                                         _ptr(self.v_xys), _ptr(self.v_conics),
                                         _ptr(self.v_colors), _ptr(self.v_opacity), st),
              "gi2d_rasterize_sum_bwd")
        if self.kind == "cholesky":
            check(lib.gi2d_project_cholesky_bwd(_ptr(means), _ptr(cov_a), self.N, self.H, self.W,
                                                _ptr(self.v_xys), _ptr(self.v_conics),
                                                _ptr(self.v_means), _ptr(self.v_cov_a), st),
                  "gi2d_project_cholesky_bwd")
        else:
            check(lib.gi2d_project_rs_bwd(_ptr(means), _ptr(cov_a), _ptr(cov_b), self.N, self.H,
                                          self.W, _ptr(self.v_xys), _ptr(self.v_conics),
                                          _ptr(self.v_means), _ptr(self.v_cov_a),
                                          _ptr(self.v_cov_b), st), "gi2d_project_rs_bwd")
        return self.v_means, self.v_cov_a, self.v_cov_b, self.v_colors, self.v_opacity

    def intersections(self):
        """Number of (tile, Gaussian) intersections of the last binning (synchronises)."""
        return int(self.ws.offsets()[-1].item())


class FrameRing:
    """``depth`` independent frames in flight for one caller: a ring of ``FramePipeline``s, each
    with its own workspace, output buffers and HIP stream (default priority).  A single frame's
    two kernels are latency-bound and leave most of the chip idle; consecutive frames of a
    sequence (a video, a batch of images) do not depend on each other, so the projection of one
    runs under the rasteriser of the one before.

    ``render(...)`` issues the next frame and returns ``(image, event)``: the image buffer
    belongs to the ring and is valid once ``event`` has completed (``event.synchronize()``, or
    ``torch.cuda.current_stream().wait_event(event)`` before consuming it on another stream);
    it is overwritten ``depth`` calls later.  Inputs are read on the ring's streams: they must
    have been produced before the call on the caller's current stream (the ring waits for that
    stream) and must not be overwritten until the returned event has completed.
    """

    def __init__(self, num_points, H, W, kind="cholesky", device="cuda:0", depth=2):
        if depth < 1:
            raise ValueError("depth >= 1 required")
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self.pipes, self.events = [], []
        for st in self.streams:
            with torch.cuda.stream(st):
                self.pipes.append(FramePipeline(num_points, H, W, kind=kind, device=self.device))
            self.events.append(torch.cuda.Event())
        self.bound = [(None, None)] * depth
        self.turn = 0

    def render(self, means, cov_a, cov_b, colors, opacity, wait_inputs=True):
        """``wait_inputs=False`` skips the wait for the caller's current stream (inputs that
        were complete long ago, e.g. a resident model rendered repeatedly)."""
        k = self.turn % len(self.pipes)
        self.turn += 1
        st = self.streams[k]
        if wait_inputs:
            st.wait_stream(torch.cuda.current_stream(self.device))  # inputs are ready there
        tensors = (means, cov_a, cov_b, colors, opacity)
        key = tuple(None if t is None else t.data_ptr() for t in tensors)
        if self.bound[k][0] != key:   # the per-frame argument list is rebuilt only for new buffers
            with torch.cuda.stream(st):
                self.bound[k] = (key, self.pipes[k].bind(*tensors))
        img = self.bound[k][1]()
        self.events[k].record(st)
        return img, self.events[k]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


class FrameBatch:
    """``frames`` independent frames of one shape rendered by ONE call and two launches
    (``gi2d_render_batch``): the projection runs over ``frames * N`` Gaussians, the rasteriser
    over ``frames * tiles`` tiles.  Inputs are contiguous ``[frames, N, ...]`` tensors; outputs
    are owned by the object: ``images`` ``[frames, H, W, 3]`` and the projection tensors
    ``[frames, N, ...]``.  Every frame is bit-identical to the one ``FramePipeline.forward``
    renders from the same inputs.  For callers with many images of one size -- the images of a
    dataset, the frames of a sequence -- where one 768x512 frame leaves most of the chip idle.
    """

    def __init__(self, frames, num_points, H, W, kind="cholesky", device="cuda:0"):
        if kind not in ("cholesky", "rs"):
            raise ValueError("kind must be 'cholesky' or 'rs'")
        if frames < 1:
            raise ValueError("frames >= 1 required")
        self.F, self.N, self.H, self.W, self.kind = int(frames), int(num_points), int(H), int(W), kind
        self.device = torch.device(device)
        self.ws = Workspace(self.device)
        self.conv = _lib.default_conv(**get_conventions())
        d, F, N = self.device, self.F, self.N
        f32, i32 = torch.float32, torch.int32
        self.xys = torch.empty(F, N, 2, dtype=f32, device=d)
        self.depths = torch.empty(F, N, dtype=f32, device=d)
        self.radii = torch.empty(F, N, dtype=i32, device=d)
        self.conics = torch.empty(F, N, 3, dtype=f32, device=d)
        self.num_tiles_hit = torch.empty(F, N, dtype=i32, device=d)
        self.images = torch.empty(F, self.H, self.W, 3, dtype=f32, device=d)
        self.background = torch.ones(3, dtype=f32, device=d)

    def _check(self, name, t, tail):
        if t is None:
            return
        want = (self.F, self.N) + tail
        if tuple(t.shape) != want or t.dtype != torch.float32 or not t.is_contiguous() or \
                t.device != self.device:
            raise ValueError(f"{name} must be a contiguous float32 {want} tensor on {self.device}")

    def bind(self, means, cov_a, cov_b, colors, opacity, out=None):
        """Checks the tensors once and returns ``run(events=None)`` for render loops."""
        out = self.images if out is None else out
        self._check("means", means, (2,))
        self._check("cov_a", cov_a, (3,) if self.kind == "cholesky" else (2,))
        if self.kind == "rs":
            if cov_b is None or cov_b.numel() != self.F * self.N:
                raise ValueError("cov_b (rotation) must have frames * N elements")
        self._check("colors", colors, (3,))
        if opacity.numel() != self.F * self.N or not opacity.is_contiguous():
            raise ValueError("opacity must be a contiguous tensor of frames * N elements")
        args = (self.ws.handle, 0 if self.kind == "cholesky" else 1, self.F, _ptr(means),
                _ptr(cov_a), _ptr(cov_b) if self.kind == "rs" else None, _ptr(colors),
                _ptr(opacity), _ptr(self.background), self.N, self.H, self.W,
                ctypes.byref(self.conv), _ptr(self.xys), _ptr(self.depths), _ptr(self.radii),
                _ptr(self.conics), _ptr(self.num_tiles_hit), _ptr(out))
        stream = _stream(self.device)
        keep = (means, cov_a, cov_b, colors, opacity, out)
        fn = lib.gi2d_render_batch

        def run(events=None, _keep=keep):
            if events is None:
                rc = fn(*args, None, None, stream)
            else:
                rc = fn(*args, events[0].cuda_event, events[1].cuda_event, stream)
            if rc:
                check(rc, "gi2d_render_batch")
            return out

        return run

    def forward(self, means, cov_a, cov_b, colors, opacity, out=None):
        return self.bind(means, cov_a, cov_b, colors, opacity, out)()

    def intersections(self):
        """(tile, Gaussian) pairs in the current lists of all frames (synchronises)."""
        return int(self.ws.offsets()[-1].item())


class BatchRing:
    """``depth`` batches in flight for one caller: a ring of ``FrameBatch`` objects, each with its
    own workspace, outputs and (default-priority) stream, used in turn.  The projection of a batch
    of 768x512 frames is bound by latency and atomics and the rasteriser by arithmetic: with two
    batches in flight the projection of one runs under the rasteriser of the other and the chip
    stays full (8 x 30 000 Gaussians at 768x512: 45 us per batch against 55 us one batch at a
    time).  The streams run free of each other -- ordering the rasterisers with events between
    the streams costs more than it saves on this part.

    ``render(...)`` issues the next batch and returns ``(images, event)``: the images belong to
    the ring slot, are valid once ``event`` has completed and are overwritten ``depth`` calls
    later.  Inputs (contiguous ``[frames, N, ...]`` tensors, as for ``FrameBatch``) are read on
    the slot's stream: they must be complete when ``render`` is called (or pass
    ``wait_inputs=True`` to order the slot behind the caller's current stream) and must stay
    unchanged until the returned event has completed.  Every image is bit-identical to
    ``FrameBatch.forward``'s.
    """

    def __init__(self, frames, num_points, H, W, kind="cholesky", device="cuda:0", depth=2):
        if depth < 1:
            raise ValueError("depth >= 1 required")
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self.slots = [FrameBatch(frames, num_points, H, W, kind=kind, device=self.device)
                      for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.bound = [(None, None)] * depth
        self.turn = 0

    def next_slot(self):
        """The ``FrameBatch`` (workspace, outputs) the next ``render`` will use."""
        return self.slots[self.turn % len(self.slots)]

    def render(self, means, cov_a, cov_b, colors, opacity, wait_inputs=False):
        k = self.turn % len(self.slots)
        self.turn += 1
        st = self.streams[k]
        if wait_inputs:
            st.wait_stream(torch.cuda.current_stream(self.device))
        tensors = (means, cov_a, cov_b, colors, opacity)
        key = tuple(None if t is None else t.data_ptr() for t in tensors)
        if self.bound[k][0] != key:   # (the argument list is rebuilt only for new buffers)
            with torch.cuda.stream(st):
                self.bound[k] = (key, self.slots[k].bind(*tensors))
        images = self.bound[k][1]()
        self.events[k].record(st)
        return images, self.events[k]

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
