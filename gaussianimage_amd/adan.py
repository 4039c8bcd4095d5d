"""Adan with the reference's constructor and update rule
(/root/reference/optimizer.py:39-101,125-235,296-362), stepped by ONE fused HIP launch over
all parameter tensors (``gi2d_adan_step``) instead of 18 multi-tensor passes.

GPU only, like the rest of the package: parameters on the CPU raise.
"""
import ctypes

import torch
from torch.optim.optimizer import Optimizer

from . import _lib
from ._lib import check, lib

__all__ = ["Adan"]


class Adan(Optimizer):
    """Constructor arguments as the reference's ``Adan`` (optimizer.py:65-74); ``foreach`` and
    ``fused`` are accepted and ignored -- there is one implementation here, the fused launch."""

    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=0.0, no_prox=False, foreach=True, fused=True):
        for name, value in (("lr", lr), ("eps", eps), ("max_grad_norm", max_grad_norm)):
            if not value >= 0.0:
                raise ValueError(f"Adan: {name} must be >= 0, got {value}")
        if len(betas) != 3 or not all(0.0 <= b < 1.0 for b in betas):
            raise ValueError(f"Adan: three betas in [0, 1) expected, got {betas}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps,
                                      weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                      no_prox=no_prox, foreach=foreach, fused=fused))

    @torch.no_grad()
    def restart_opt(self):
        """Back to step 0 with zeroed moments (optimizer.py:108-122)."""
        for group in self.param_groups:
            group['step'] = 0
        for state in self.state.values():
            for key in ('exp_avg', 'exp_avg_sq', 'exp_avg_diff'):
                if key in state:
                    state[key].zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip = None
        if self.defaults['max_grad_norm'] > 0:
            # same expression as the reference, kept on the device (no .item() synchronisation)
            grads = [p.grad for g in self.param_groups for p in g['params'] if p.grad is not None]
            if grads:
                norm = torch.sqrt(sum(g.float().pow(2).sum() for g in grads))
                eps = self.param_groups[-1]['eps']
                clip = torch.clamp(self.defaults['max_grad_norm'] / (norm + eps),
                                   max=1.0).to(torch.float32).reshape(1).contiguous()
        for group in self.param_groups:
            group['step'] = group.get('step', 0) + 1
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            cols = ([], [], [], [], [], [])
            for p in ps:
                if not p.is_cuda:
                    raise RuntimeError("gaussianimage_amd.Adan steps parameters on an AMD GPU "
                                       "only (there is no CPU fallback)")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("Adan: parameters must be contiguous float32")
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    p.grad = p.grad.to(torch.float32).contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                    state['exp_avg_diff'] = torch.zeros_like(p)
                if 'neg_pre_grad' not in state:
                    state['neg_pre_grad'] = torch.empty_like(p)
                    fresh = True
                else:
                    fresh = False
                # a parameter that joins late starts like step 1 (optimizer.py:187-189)
                if fresh and group['step'] != 1:
                    state['neg_pre_grad'].copy_(p.grad).mul_(-1.0 if clip is None else -clip)
                for c, t in zip(cols, (p, p.grad, state['exp_avg'], state['exp_avg_sq'],
                                       state['exp_avg_diff'], state['neg_pre_grad'])):
                    c.append(t.data_ptr())
            n = len(ps)
            arrs = [(ctypes.c_void_p * n)(*c) for c in cols]
            numel = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
            beta1, beta2, beta3 = group['betas']
            hyper = _lib.AdanHyper(group['lr'], beta1, beta2, beta3, group['eps'],
                                   group['weight_decay'], int(bool(group['no_prox'])),
                                   int(group['step']))
            dev = ps[0].device
            with torch.cuda.device(dev):
                st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                check(lib.gi2d_adan_step(n, *arrs, numel, ctypes.byref(hyper),
                                         ctypes.c_void_p(clip.data_ptr()) if clip is not None
                                         else None, st), "gi2d_adan_step")
        return loss
