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

__all__ = ["band_plan", "band_pixels", "BandGather", "all_reduce_grads", "ShardedFrame",
           "bench_tile_row_sharded"]


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
            "allreduce_bytes": 8 * N * 4}
