"""Fused fitting loop: the reference's ``train_iter`` (forward, L2 loss, backward, Adan step --
/root/reference/gaussianimage_cholesky.py:101-124, /root/reference/gaussianimage_rs.py:82-95)
in two launches on the model's raw parameters, with no gradient tensors and no host
synchronisation (``gi2d_fit_step``: per tile the rasteriser with the loss epilogue followed at
once by the tile's backward pass; per Gaussian the gradient gather, the Adan update and, in the
same lane, the projection of the updated parameters for the next iteration).  The three- and
four-launch forms (``gi2d_fit_forward`` + ``gi2d_fit_backward_step[_chain]``) stay available
through ``one_pass=False`` / ``chain=False`` and give the same bits.

This is an optional fast path beside the drop-in operators (which the reference's own model
files keep using unchanged).  Parameter names and shapes are the reference model's
(``_xyz``, ``_cholesky`` / ``_scaling`` + ``_rotation``, ``_features_dc``, buffer ``_opacity``),
so a ``state_dict`` moves between the two.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import BLOCK, check, lib
from .ops import Workspace, _ptr, _stream, get_conventions

__all__ = ["FusedFitter", "BatchFitter"]


class FusedFitter:
    def __init__(self, num_points, H, W, kind="cholesky", device="cuda:0", lr=1e-3,
                 betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, no_prox=False,
                 lr_step=20000, lr_gamma=0.5, params=None, keep_image=True, chain=True,
                 one_pass=True, spatial_order=False):
        if kind not in ("cholesky", "rs"):
            raise ValueError("kind must be 'cholesky' or 'rs'")
        self.kind, self.N, self.H, self.W = kind, int(num_points), int(H), int(W)
        self.device = torch.device(device)
        self.lr0, self.betas, self.eps = lr, betas, eps
        self.weight_decay, self.no_prox = weight_decay, no_prox
        self.lr_step, self.lr_gamma = lr_step, lr_gamma
        N, d, f32 = self.N, self.device, torch.float32
        if params is None:  # the reference's initial distributions
            params = {"_xyz": torch.atanh(2 * (torch.rand(N, 2) - 0.5)),
                      "_features_dc": torch.rand(N, 3)}
            if kind == "cholesky":
                params["_cholesky"] = torch.rand(N, 3)
            else:
                params["_scaling"] = torch.rand(N, 2)
                params["_rotation"] = torch.rand(N, 1)
        names = (["_xyz", "_cholesky", None, "_features_dc"] if kind == "cholesky" else
                 ["_xyz", "_scaling", "_rotation", "_features_dc"])
        self.names = names
        self.params = {k: params[k].detach().to(d, f32).contiguous().clone()
                       for k in names if k}
        if spatial_order:
            # Gaussians stored in the order of the tiles their (initial) centres fall in: the
            # lanes of a wave then enter the same few tiles and share their list atomics.  The
            # model does not depend on the order of its Gaussians; positions move little while
            # fitting, so the order is set once.
            with torch.no_grad():
                m = torch.tanh(self.params["_xyz"])
                col = ((0.5 * W * m[:, 0] + 0.5 * W) / BLOCK).floor().clamp_(0, (W - 1) // BLOCK)
                row = ((0.5 * H * m[:, 1] + 0.5 * H) / BLOCK).floor().clamp_(0, (H - 1) // BLOCK)
                order = torch.argsort(row * ((W + BLOCK - 1) // BLOCK) + col, stable=True)
                for k in self.params:
                    self.params[k] = self.params[k][order].contiguous()
        self._opacity = torch.ones(N, 1, dtype=f32, device=d)
        self.background = torch.ones(3, dtype=f32, device=d)  # gaussianimage_cholesky.py:38
        self.state = {k: {s: torch.zeros_like(p) for s in
                          ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad")}
                      for k, p in self.params.items()}
        self.step_count = 0
        # chain: the step's last kernel also projects the updated parameters for the next
        # iteration (one launch fewer).  The parameter tensors' versions tell whether anything
        # but the fused step touched them in between; if so the next forward projects afresh.
        self.chain = bool(chain)
        # one_pass: forward and backward rasterisation in one kernel (gi2d_fit_step, two launches
        # per iteration; implies the chained tail).  False keeps the separate calls.
        self.one_pass = bool(one_pass) and self.chain
        self._versions = None
        self.overflowed_frames = 0  # iterations in which some tile list overflowed (see forward)
        self.ws = Workspace(d)
        self.conv = _lib.default_conv(**get_conventions())
        self.tiles = ((W + BLOCK - 1) // BLOCK) * ((H + BLOCK - 1) // BLOCK)
        i32 = torch.int32
        self.xys = torch.empty(N, 2, dtype=f32, device=d)
        self.depths = torch.empty(N, dtype=f32, device=d)
        self.radii = torch.empty(N, dtype=i32, device=d)
        self.conics = torch.empty(N, 3, dtype=f32, device=d)
        self.num_tiles_hit = torch.empty(N, dtype=i32, device=d)
        self.image = torch.empty(H, W, 3, dtype=f32, device=d) if keep_image else None
        self.v_image = torch.empty(H, W, 3, dtype=f32, device=d)
        self.loss_partials = torch.zeros(self.tiles, dtype=f32, device=d)
        self._bind()

    def _bind(self):
        p = [self.params[k] if k else None for k in self.names]
        self._kind = 0 if self.kind == "cholesky" else 1
        self._fwd_args = (self.ws.handle, self._kind, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]),
                          _ptr(p[3]), _ptr(self._opacity), _ptr(self.background))
        self._fwd_tail = (self.N, self.H, self.W, ctypes.byref(self.conv), _ptr(self.xys),
                          _ptr(self.depths), _ptr(self.radii), _ptr(self.conics),
                          _ptr(self.num_tiles_hit), _ptr(self.image), _ptr(self.v_image),
                          _ptr(self.loss_partials))
        arr = lambda key: (ctypes.c_void_p * 4)(*[  # noqa: E731
            self.state[k][key].data_ptr() if k else None for k in self.names])
        self._state_arrays = tuple(arr(k) for k in
                                   ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"))
        self._bwd_args = (self.ws.handle, self._kind, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]),
                          _ptr(p[3]), _ptr(self._opacity), self.N, self.H, self.W,
                          ctypes.byref(self.conv), _ptr(self.xys), _ptr(self.radii),
                          _ptr(self.conics), _ptr(self.v_image), *self._state_arrays)
        self._chain_args = (self.ws.handle, self._kind, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]),
                            _ptr(p[3]), _ptr(self._opacity), self.N, self.H, self.W,
                            ctypes.byref(self.conv), _ptr(self.xys), _ptr(self.depths),
                            _ptr(self.radii), _ptr(self.conics), _ptr(self.num_tiles_hit),
                            _ptr(self.v_image), *self._state_arrays)
        self._step_head = (self.ws.handle, self._kind, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]),
                           _ptr(p[3]), _ptr(self._opacity), _ptr(self.background))
        self._step_tail = (self.N, self.H, self.W, ctypes.byref(self.conv), _ptr(self.xys),
                           _ptr(self.depths), _ptr(self.radii), _ptr(self.conics),
                           _ptr(self.num_tiles_hit), _ptr(self.image), _ptr(self.loss_partials),
                           *self._state_arrays)
        self._stream = _stream(self.device)

    @property
    def lr(self):
        """StepLR(step_size, gamma) of the reference (gaussianimage_cholesky.py:55,123)."""
        return self.lr0 * self.lr_gamma ** (self.step_count // self.lr_step)

    def _check_target(self, gt_image):
        if gt_image.device != self.device or gt_image.dtype != torch.float32 or \
                not gt_image.is_contiguous() or gt_image.numel() != 3 * self.H * self.W:
            raise ValueError("gt_image must be a contiguous float32 [1,3,H,W] tensor on the device")
        if self._versions is not None:
            if self._versions != self._param_versions():
                # someone else wrote the parameters since the chained step projected them
                check(lib.gi2d_workspace_set_option(self.ws.handle, _lib.OPT_DROP_PENDING, 1),
                      "gi2d_workspace_set_option")
            self._versions = None

    def _late_overflow(self, rc):
        if rc == _lib.GI2D_E_PREV_OVERFLOW:
            # An EARLIER iteration outgrew a tile list (Gaussians grow while fitting): its
            # overflowed tiles contributed no loss gradient for that one step, the lists have
            # been enlarged since, and this iteration is complete.  Nothing to repair in a fit.
            self.overflowed_frames += 1
            rc = 0
        return rc

    def forward(self, gt_image):
        """Renders and evaluates the loss gradient; ``gt_image``: [1,3,H,W] or [3,H,W] float32."""
        self._check_target(gt_image)
        rc = lib.gi2d_fit_forward(*self._fwd_args, _ptr(gt_image), *self._fwd_tail, self._stream)
        check(self._late_overflow(rc), "gi2d_fit_forward")

    def step(self, gt_image):
        """One fitting iteration (forward, loss, backward, Adan).  Returns nothing: read
        ``loss()`` / ``psnr()`` when needed -- they synchronise, the step does not."""
        lr = self.lr
        b1, b2, b3 = self.betas
        if self.one_pass:
            self._check_target(gt_image)
            self.step_count += 1
            hyper = _lib.AdanHyper(lr, b1, b2, b3, self.eps, self.weight_decay,
                                   int(bool(self.no_prox)), self.step_count)
            rc = lib.gi2d_fit_step(*self._step_head, _ptr(gt_image), *self._step_tail,
                                   ctypes.byref(hyper), self._stream)
            self._versions = self._param_versions()
            check(self._late_overflow(rc), "gi2d_fit_step")
            return
        self.forward(gt_image)
        self.step_count += 1
        hyper = _lib.AdanHyper(lr, b1, b2, b3, self.eps, self.weight_decay,
                               int(bool(self.no_prox)), self.step_count)
        if self.chain:
            rc = lib.gi2d_fit_backward_step_chain(*self._chain_args, ctypes.byref(hyper),
                                                  self._stream)
            self._versions = self._param_versions()
        else:
            rc = lib.gi2d_fit_backward_step(*self._bwd_args, ctypes.byref(hyper), self._stream)
        check(rc, "gi2d_fit_backward_step")

    def _param_versions(self):
        return tuple(p._version for p in self.params.values())

    def loss(self):
        """Mean squared error of the last forward (before that step's update)."""
        return float(self.loss_partials.double().sum().item()) / (3 * self.H * self.W)

    def psnr(self):
        return 10 * math.log10(1.0 / max(self.loss(), 1e-30))

    def render(self):
        """The last forward's image as the model returns it: [1,3,H,W], clamped to [0,1]."""
        if self.image is None:
            raise RuntimeError("created with keep_image=False")
        return torch.clamp(self.image, 0, 1).view(-1, self.H, self.W, 3).permute(
            0, 3, 1, 2).contiguous()

    def state_dict(self):
        sd = {k: v.clone() for k, v in self.params.items()}
        sd["_opacity"] = self._opacity.clone()
        return sd


class BatchFitter:
    """``frames`` images of one size fitted in lockstep by ``gi2d_fit_step_batch``: the fused
    iteration of ``FusedFitter`` over all of them in one call (three launches from four images
    on).  Parameters, optimizer
    state and outputs are ``[frames, N, ...]`` tensors (``params["_xyz"][f]`` is image f's
    ``_xyz``); targets are one contiguous ``[frames, 3, H, W]`` tensor.  Image f follows the
    trajectory ``FusedFitter`` gives it alone -- bit for bit when both run the same workgroup
    shape (``Workspace.set_raster_ppl``).  Left to themselves they may not: a batch of four or
    more images projects through the LDS-aggregated kernel and, from 20 480 tiles per launch,
    rasterises with one wave per tile where a single 768x512 image takes four; the backward then
    adds a Gaussian's per-pixel terms in another order, and the last-bit gradient differences
    grow to ~6e-4 of a parameter within a few Adan steps.  For the reference's loop over the
    images of a dataset (/root/reference/train.py:366-372), where one 768x512 fit leaves most of
    the chip idle.
    """

    def __init__(self, frames, num_points, H, W, kind="cholesky", device="cuda:0", lr=1e-3,
                 betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, no_prox=False,
                 lr_step=20000, lr_gamma=0.5, params=None, keep_images=False):
        if kind not in ("cholesky", "rs"):
            raise ValueError("kind must be 'cholesky' or 'rs'")
        if frames < 1:
            raise ValueError("frames >= 1 required")
        self.kind, self.F, self.N = kind, int(frames), int(num_points)
        self.H, self.W = int(H), int(W)
        self.device = torch.device(device)
        self.lr0, self.betas, self.eps = lr, betas, eps
        self.weight_decay, self.no_prox = weight_decay, no_prox
        self.lr_step, self.lr_gamma = lr_step, lr_gamma
        F, N, d, f32, i32 = self.F, self.N, self.device, torch.float32, torch.int32
        names = (["_xyz", "_cholesky", None, "_features_dc"] if kind == "cholesky" else
                 ["_xyz", "_scaling", "_rotation", "_features_dc"])
        dims = {"_xyz": 2, "_cholesky": 3, "_scaling": 2, "_rotation": 1, "_features_dc": 3}
        if params is None:  # the reference's initial distributions, image by image
            params = {k: (torch.atanh(2 * (torch.rand(F, N, 2) - 0.5)) if k == "_xyz"
                          else torch.rand(F, N, dims[k])) for k in names if k}
        self.names = names
        self.params = {}
        for k in names:
            if k:
                p = params[k].detach().to(d, f32).contiguous().clone()
                if tuple(p.shape) != (F, N, dims[k]):
                    raise ValueError(f"{k} must have shape {(F, N, dims[k])}")
                self.params[k] = p
        self._opacity = torch.ones(F, N, 1, dtype=f32, device=d)
        self.background = torch.ones(3, dtype=f32, device=d)
        self.state = {k: {s: torch.zeros_like(p) for s in
                          ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad")}
                      for k, p in self.params.items()}
        self.step_count = 0
        self.overflowed_frames = 0
        self.ws = Workspace(d)
        self.conv = _lib.default_conv(**get_conventions())
        self.tiles = ((W + BLOCK - 1) // BLOCK) * ((H + BLOCK - 1) // BLOCK)
        self.xys = torch.empty(F, N, 2, dtype=f32, device=d)
        self.depths = torch.empty(F, N, dtype=f32, device=d)
        self.radii = torch.empty(F, N, dtype=i32, device=d)
        self.conics = torch.empty(F, N, 3, dtype=f32, device=d)
        self.num_tiles_hit = torch.empty(F, N, dtype=i32, device=d)
        self.images = torch.empty(F, H, W, 3, dtype=f32, device=d) if keep_images else None
        self.loss_partials = torch.zeros(F, self.tiles, dtype=f32, device=d)
        self._versions = None
        p = [self.params[k] if k else None for k in names]
        arr = lambda key: (ctypes.c_void_p * 4)(*[  # noqa: E731
            self.state[k][key].data_ptr() if k else None for k in names])
        self._state_arrays = tuple(arr(k) for k in
                                   ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"))
        self._head = (self.ws.handle, 0 if kind == "cholesky" else 1, F, _ptr(p[0]), _ptr(p[1]),
                      _ptr(p[2]), _ptr(p[3]), _ptr(self._opacity), _ptr(self.background))
        self._tail = (N, self.H, self.W, ctypes.byref(self.conv), _ptr(self.xys),
                      _ptr(self.depths), _ptr(self.radii), _ptr(self.conics),
                      _ptr(self.num_tiles_hit), _ptr(self.images), _ptr(self.loss_partials),
                      *self._state_arrays)
        self._stream = _stream(self.device)

    @property
    def lr(self):
        return self.lr0 * self.lr_gamma ** (self.step_count // self.lr_step)

    def step(self, gt_images):
        """One fitting iteration of every image; ``gt_images``: contiguous float32
        ``[frames, 3, H, W]`` on the device.  No host synchronisation."""
        if gt_images.device != self.device or gt_images.dtype != torch.float32 or \
                not gt_images.is_contiguous() or \
                gt_images.numel() != self.F * 3 * self.H * self.W:
            raise ValueError("gt_images must be a contiguous float32 [frames,3,H,W] tensor on "
                             "the device")
        versions = tuple(p._version for p in self.params.values())
        if self._versions is not None and self._versions != versions:
            # someone else wrote the parameters since the chained step projected them
            check(lib.gi2d_workspace_set_option(self.ws.handle, _lib.OPT_DROP_PENDING, 1),
                  "gi2d_workspace_set_option")
        b1, b2, b3 = self.betas
        lr = self.lr
        self.step_count += 1
        hyper = _lib.AdanHyper(lr, b1, b2, b3, self.eps, self.weight_decay,
                               int(bool(self.no_prox)), self.step_count)
        rc = lib.gi2d_fit_step_batch(*self._head, _ptr(gt_images), *self._tail,
                                     ctypes.byref(hyper), self._stream)
        self._versions = tuple(p._version for p in self.params.values())
        if rc == _lib.GI2D_E_PREV_OVERFLOW:   # (see FusedFitter._late_overflow)
            self.overflowed_frames += 1
            rc = 0
        check(rc, "gi2d_fit_step_batch")

    def losses(self):
        """Mean squared error of every image in the last step's forward (synchronises)."""
        return (self.loss_partials.double().sum(dim=1) / (3 * self.H * self.W)).cpu()

    def psnrs(self):
        return [10 * math.log10(1.0 / max(float(v), 1e-30)) for v in self.losses()]

    def state_dict(self, frame):
        """Image ``frame``'s parameters under the reference model's names."""
        sd = {k: v[frame].clone() for k, v in self.params.items()}
        sd["_opacity"] = self._opacity[frame].clone()
        return sd
