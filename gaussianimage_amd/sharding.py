"""Tile-row sharding of one frame across the GPUs of a node (one process per GPU).

Pixels of different tiles are independent in the forward pass, and the Gaussians are small
(8 floats each) and replicated on every rank.  Rank ``r`` of ``R`` owns a contiguous band of
tile rows, projects all Gaussians, bins and rasterises only its band
(``gi2d_workspace_set_rows``), and the bands are concatenated by ONE all-gather (RCCL over
xGMI when the process group's backend is ``nccl``).  In the backward pass the gradient of the
image is sliced by band, every rank produces partial parameter gradients from its band, and
those are summed with one all-reduce.

The reference is single-GPU (/root/reference/scripts/gaussianimage_cholesky/kodak.sh:13 pins
``CUDA_VISIBLE_DEVICES=0``); this module has no counterpart there.

The exchange functions accept CPU tensors with a ``gloo`` group (that is how the CPU tests run
them); device tensors with a ``gloo`` group are staged through host memory.
"""
import time

import torch
import torch.distributed as dist

from ._lib import BLOCK

__all__ = ["band_plan", "band_pixels", "BandGather", "all_reduce_grads", "GradExchange",
           "ShardedFrame", "ShardedFitter", "bench_tile_row_sharded"]


def band_plan(tiles_y, world_size):
    """Tile rows ``[begin, end)`` of every rank.  All ranks but the last get
    ``ceil(tiles_y / world_size)`` rows, so that equal-size gather slots laid end to end are
    the image itself (the last slot carries the padding); trailing ranks may be empty."""
    if tiles_y < 0 or world_size < 1:
        raise ValueError("tiles_y >= 0 and world_size >= 1 required")
    rm = -(-tiles_y // world_size) if tiles_y else 0
    return [(min(r * rm, tiles_y), min((r + 1) * rm, tiles_y)) for r in range(world_size)]


def band_pixels(plan, rank, H):
    """Pixel rows ``[y0, y1)`` of a rank's band."""
    r0, r1 = plan[rank]
    return min(r0 * BLOCK, H), min(r1 * BLOCK, H)


def _backend(group):
    return dist.get_backend(group)


class BandGather:
    """Reassembles the frame from per-rank bands with one all-gather.

    Every rank contributes a slot of ``rows_max*16`` pixel rows (its band, zero padded); the
    concatenation of the slots, cut at ``H`` rows, is the image.  Slot buffers are allocated
    once; ``depth`` of them rotate so that a gather may overlap the next frame's rendering.
    """

    def __init__(self, H, W, world_size, rank, device, channels=3, group=None, depth=2):
        self.H, self.W, self.C = H, W, channels
        self.world, self.rank, self.group = world_size, rank, group
        self.tiles_y = (H + BLOCK - 1) // BLOCK
        self.plan = band_plan(self.tiles_y, world_size)
        self.rows_max = max(e - b for b, e in self.plan) if self.plan else 0
        self.slot_rows = self.rows_max * BLOCK
        self.device = torch.device(device)
        self.slots = [torch.zeros(self.slot_rows, W, channels, dtype=torch.float32,
                                  device=self.device) for _ in range(depth)]
        self.frames = [torch.empty(world_size * self.slot_rows, W, channels, dtype=torch.float32,
                                   device=self.device) for _ in range(depth)]
        self.turn = 0

    def band_view(self, k=None):
        """``[band_h, W, C]`` view of the current slot: render the band straight into it."""
        k = self.turn if k is None else k
        y0, y1 = band_pixels(self.plan, self.rank, self.H)
        return self.slots[k % len(self.slots)][: y1 - y0]

    def gather(self, async_op=False):
        """All-gathers the current slot; returns ``(image[H,W,C] view, work or None)`` and
        rotates to the next slot."""
        k = self.turn % len(self.slots)
        slot, frame = self.slots[k], self.frames[k]
        self.turn += 1
        if self.world == 1:
            frame.copy_(slot)
            return frame[: self.H], None
        if _backend(self.group) == "gloo" and slot.is_cuda:
            # gloo gets host tensors only
            h_slot = slot.cpu()
            h_frame = torch.empty(frame.shape, dtype=frame.dtype)
            dist.all_gather_into_tensor(h_frame, h_slot, group=self.group)
            frame.copy_(h_frame)
            return frame[: self.H], None
        work = dist.all_gather_into_tensor(frame, slot, group=self.group, async_op=async_op)
        return frame[: self.H], work


    def gather_to(self, dst=0):
        """Collects the bands on ONE rank (a viewer, an encoder, a loss that lives on one GPU):
        every rank sends its slot straight to ``dst`` -- point-to-point, so the seven xGMI links
        into that GPU carry one band each at the same time, where an all-gather delivers the
        whole frame to every rank.  Returns the ``[H,W,C]`` image view on ``dst`` and ``None``
        elsewhere; rotates to the next slot."""
        k = self.turn % len(self.slots)
        slot, frame = self.slots[k], self.frames[k]
        self.turn += 1
        if self.world == 1:
            frame.copy_(slot)
            return frame[: self.H]
        staged = _backend(self.group) == "gloo" and slot.is_cuda  # gloo gets host tensors only
        src = slot.cpu() if staged else slot
        if self.rank != dst:
            dist.gather(src, dst=dst, group=self.group)
            return None
        dst_frame = torch.empty(frame.shape, dtype=frame.dtype) if staged else frame
        parts = [dst_frame[r * self.slot_rows:(r + 1) * self.slot_rows] for r in range(self.world)]
        dist.gather(src, gather_list=parts, dst=dst, group=self.group)
        if staged:
            frame.copy_(dst_frame)
        return frame[: self.H]


def all_reduce_grads(grads, group=None):
    """Sums the per-band partial parameter gradients over the ranks, in place, with a single
    collective over one flat buffer (``grads``: list of same-dtype tensors)."""
    grads = [g for g in grads if g is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return grads
    flat = torch.cat([g.reshape(-1) for g in grads])
    staged = _backend(group) == "gloo" and flat.is_cuda
    buf = flat.cpu() if staged else flat
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if staged:
        flat.copy_(buf)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return grads


class ShardedFrame:
    """One frame configuration rendered by ``world_size`` ranks, split by tile rows."""

    def __init__(self, num_points, H, W, kind, rank, world_size, device, group=None):
        from .pipeline import FramePipeline
        self.gather = BandGather(H, W, world_size, rank, device, group=group)
        self.rank, self.world, self.group = rank, world_size, group
        self.pipe = FramePipeline(num_points, H, W, kind=kind, device=device,
                                  rows=self.gather.plan[rank])
        self.H, self.W = H, W

    def render_band(self, means, cov_a, cov_b, colors, opacity):
        """Renders this rank's band into the current gather slot."""
        band = self.gather.band_view()
        return self.pipe.forward(means, cov_a, cov_b, colors, opacity, out=band)

    def render(self, means, cov_a, cov_b, colors, opacity, async_op=False):
        """Band render plus the all-gather; returns ``(image[H,W,3], work)``."""
        self.render_band(means, cov_a, cov_b, colors, opacity)
        return self.gather.gather(async_op=async_op)

    def band_grad(self, v_image):
        """This rank's rows of a full ``[H,W,3]`` image gradient (a contiguous view)."""
        y0, y1 = band_pixels(self.gather.plan, self.rank, self.H)
        return v_image[y0:y1]

    def backward(self, means, cov_a, cov_b, colors, opacity, v_image):
        """``v_image`` is the full ``[H,W,3]`` image gradient (every rank has it, or at least
        its own band of it); returns the summed parameter gradients on every rank."""
        grads = self.pipe.backward(means, cov_a, cov_b, colors, opacity,
                                   self.band_grad(v_image))
        all_reduce_grads(list(grads), self.group)
        return grads


class GradExchange:
    """The one exchange of a tile-row sharded fitting step: every rank's per-Gaussian partial
    gradients of the rasteriser -- v_xys[N,2], v_conics[N,3], v_colors[N,3], v_opacity[N], views
    of ONE flat buffer of 9 N floats -- are summed over the ranks by a single all-reduce (RCCL
    when the backend is ``nccl``; a ``gloo`` group gets the buffer through host memory)."""

    def __init__(self, num_points, device, group=None):
        N = int(num_points)
        self.group = group
        self.flat = torch.zeros(9 * N, dtype=torch.float32, device=device)
        self.v_xys = self.flat[: 2 * N].view(N, 2)
        self.v_conics = self.flat[2 * N: 5 * N].view(N, 3)
        self.v_colors = self.flat[5 * N: 8 * N].view(N, 3)
        self.v_opacity = self.flat[8 * N:]

    def reduce(self):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        if _backend(self.group) == "gloo" and self.flat.is_cuda:
            buf = self.flat.cpu()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(buf)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)


class ShardedFitter:
    """The fused fitting loop with the image split by tile rows over the ranks.

    Every rank holds ALL parameters and the whole optimizer state (8 floats per Gaussian and
    their moments: a few MB) and the whole target image, and rasterises only its band of tile
    rows.  One iteration is ``gi2d_fit_raster`` on the band (projection, forward, L2 loss,
    backward partial sums, their sum per Gaussian) -> ONE all-reduce of 9 N floats ->
    ``gi2d_fit_tail`` (chain rule, Adan, and the projection of the updated parameters into the
    band's next lists): three launches and one collective, and because every rank applies the
    same update to the same parameters, the replicas never drift apart.  With one rank it gives
    the bits of ``FusedFitter``.  The reference fits on one GPU
    (/root/reference/gaussianimage_cholesky.py:101-124); this has no counterpart there."""

    def __init__(self, num_points, H, W, kind, rank, world_size, device, group=None, rows=None,
                 **fitter_args):
        import ctypes
        from . import _lib
        from ._lib import lib
        from .fit import FusedFitter
        self._ctypes, self._lib, self._clib = ctypes, _lib, lib
        self.rank, self.world, self.group = rank, world_size, group
        self.H, self.W = int(H), int(W)
        tiles_y = (self.H + BLOCK - 1) // BLOCK
        self.plan = band_plan(tiles_y, world_size)
        self.rows = self.plan[rank] if rows is None else rows
        self.fit = FusedFitter(num_points, H, W, kind=kind, device=device, keep_image=False,
                               **fitter_args)
        self.fit.ws.set_rows(*self.rows)
        tiles_x = (self.W + BLOCK - 1) // BLOCK
        # (one loss partial per tile of the band; the full-frame buffer is large enough)
        self.band_tiles = tiles_x * (self.rows[1] - self.rows[0])
        self.exchange = GradExchange(num_points, self.fit.device, group)
        self._bind()

    def _bind(self):
        from .ops import _ptr
        f, x = self.fit, self.exchange
        p = [f.params[k] if k else None for k in f.names]
        self._raster_args = (f.ws.handle, f._kind, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]), _ptr(p[3]),
                             _ptr(f._opacity), _ptr(f.background))
        self._raster_tail = (f.N, f.H, f.W, self._ctypes.byref(f.conv), _ptr(f.xys),
                             _ptr(f.depths), _ptr(f.radii), _ptr(f.conics), _ptr(f.num_tiles_hit),
                             None, _ptr(f.loss_partials), _ptr(x.v_xys), _ptr(x.v_conics),
                             _ptr(x.v_colors), _ptr(x.v_opacity))
        self._tail_args = (f.ws.handle, f._kind, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]), _ptr(p[3]),
                           _ptr(f._opacity), f.N, f.H, f.W, self._ctypes.byref(f.conv),
                           _ptr(f.xys), _ptr(f.depths), _ptr(f.radii), _ptr(f.conics),
                           _ptr(f.num_tiles_hit), _ptr(x.v_xys), _ptr(x.v_conics),
                           _ptr(x.v_colors), *f._state_arrays)

    def step(self, gt_image):
        """One fitting iteration; ``gt_image`` is the whole [1,3,H,W] target on every rank."""
        from .ops import _ptr
        f, lib, _lib = self.fit, self._clib, self._lib
        f._check_target(gt_image)
        if f.ws.rows != self.rows:
            f.ws.set_rows(*self.rows)
        # The two kernels run on the stream the fitter was bound to, the all-reduce on torch's
        # current stream: they are ordered only when these are the same stream.
        from .ops import _stream
        if _stream(f.device) != f._stream:
            raise RuntimeError("ShardedFitter.step must be called on the stream the fitter was "
                               "created on (its kernels and the all-reduce are ordered by "
                               "running on one stream)")
        lr = f.lr
        rc = lib.gi2d_fit_raster(*self._raster_args, _ptr(gt_image), *self._raster_tail, f._stream)
        _lib.check(f._late_overflow(rc), "gi2d_fit_raster")
        self.exchange.reduce()
        f.step_count += 1
        b1, b2, b3 = f.betas
        hyper = _lib.AdanHyper(lr, b1, b2, b3, f.eps, f.weight_decay, int(bool(f.no_prox)),
                               f.step_count)
        rc = lib.gi2d_fit_tail(*self._tail_args, self._ctypes.byref(hyper), f._stream)
        f._versions = f._param_versions()
        _lib.check(rc, "gi2d_fit_tail")

    def loss(self):
        """Mean squared error of the last forward over the WHOLE frame (sums the bands' partial
        sums over the ranks; synchronises)."""
        part = self.fit.loss_partials[: self.band_tiles].double().sum().reshape(1)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            buf = part.cpu() if _backend(self.group) == "gloo" else part
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            part = buf
        return float(part.item()) / (3 * self.H * self.W)

    @property
    def params(self):
        return self.fit.params


def bench_tile_row_sharded(device, rank, world, shape, inputs, steps=50, warmup=5, label=""):
    """Times one frame of ``shape = (H, W, N, kind)`` split by tile rows over ``world`` ranks
    (``inputs``: the replicated operator inputs on ``device``), max over ranks:

    * the same frame on ONE GPU (every rank renders it whole): the reference point;
    * band rendering alone -- what a consumer that keeps the bands resident on their GPUs pays
      (a per-band loss, an encoder that works per band): the figure tile-row scaling is about;
    * band rendering plus the all-gather of the frame to every rank, and overlapped with the
      next frame's rendering; plus the gather to rank 0 alone;
    * the training direction with resident bands: band forward + band backward, then ONE
      all-reduce of the 8 floats per Gaussian of parameter gradients.
    """
    from .pipeline import FramePipeline
    H, W, N, kind = shape
    frame = ShardedFrame(N, H, W, kind, rank, world, device)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                         device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps

    v_image = torch.randn(H, W, 3, device=device)
    whole = FramePipeline(N, H, W, kind=kind, device=device)
    t_one = timed(lambda: whole.forward(*inputs))
    t_one_fb = timed(lambda: (whole.forward(*inputs), whole.backward(*inputs, v_image)))
    del whole
    t_compute = timed(lambda: frame.render_band(*inputs))
    t_total = timed(lambda: frame.render(*inputs))
    pending = []

    def overlapped():
        _, work = frame.render(*inputs, async_op=True)
        pending.append(work)
        if len(pending) > 1:
            w = pending.pop(0)
            if w is not None:
                w.wait()

    t_overlap = timed(overlapped)
    for w in pending:
        if w is not None:
            w.wait()
    torch.cuda.synchronize()
    t_fb = timed(lambda: (frame.render_band(*inputs),
                          frame.pipe.backward(*inputs, frame.band_grad(v_image))))
    t_fb_ar = timed(lambda: (frame.render_band(*inputs), frame.backward(*inputs, v_image)))
    # Last, so that nothing above depends on it: the frame collected on rank 0 alone
    # (point-to-point sends instead of the all-gather).
    try:
        t_to0 = timed(lambda: (frame.render_band(*inputs), frame.gather.gather_to(0)))
    except Exception:  # an exchange this backend cannot do must not cost the other figures
        t_to0 = None
    # the fused fitting step: whole frame on one GPU (two launches) against the sharded step
    # (band rasterisation, one all-reduce of 9 N floats, tail: three launches + one collective)
    from ._testing import synth
    from .fit import FusedFitter
    gt = synth.target_image(H, W).to(device)
    torch.manual_seed(1)
    one = FusedFitter(N, H, W, kind=kind, device=device, keep_image=False)
    t_fit_one = timed(lambda: one.step(gt))
    del one
    torch.manual_seed(1)
    many = ShardedFitter(N, H, W, kind, rank, world, device)
    t_fit_many = timed(lambda: many.step(gt))
    del many
    ms = lambda t: None if t is None else t * 1e3  # noqa: E731
    return {"workload": f"{label}: {W}x{H}, {N} Gaussians, tile rows split over {world} ranks",
            "scaling": "strong", "band_rows": frame.gather.plan,
            "one_gpu_forward_ms": ms(t_one), "one_gpu_fwd_bwd_ms": ms(t_one_fb),
            "resident_bands_forward_ms": ms(t_compute),
            "resident_bands_speedup": t_one / t_compute,
            "resident_bands_efficiency": t_one / t_compute / world,
            "forward_plus_allgather_ms": ms(t_total), "allgather_speedup": t_one / t_total,
            "forward_overlapped_allgather_ms": ms(t_overlap),
            "overlapped_allgather_speedup": t_one / t_overlap,
            "forward_plus_gather_to_rank0_ms": ms(t_to0),
            "allgather_bytes_per_rank": frame.gather.slot_rows * W * 3 * 4,
            "resident_bands_fwd_bwd_ms": ms(t_fb),
            "resident_bands_fwd_bwd_plus_allreduce_ms": ms(t_fb_ar),
            "fwd_bwd_allreduce_speedup": t_one_fb / t_fb_ar,
            "fwd_bwd_allreduce_efficiency": t_one_fb / t_fb_ar / world,
            "allreduce_bytes": 8 * N * 4,
            "one_gpu_fused_fit_step_ms": ms(t_fit_one), "sharded_fit_step_ms": ms(t_fit_many),
            "sharded_fit_speedup": t_fit_one / t_fit_many,
            "sharded_fit_efficiency": t_fit_one / t_fit_many / world,
            "sharded_fit_allreduce_bytes": 9 * N * 4}
