"""numpy restatement of the reference's Adan update (/root/reference/optimizer.py:125-235 for
the step bookkeeping, :296-362 for the per-element rule), fp32 like the reference.

TEST INFRASTRUCTURE ONLY.  Parity PINNED: tests/test_adan.py checks this restatement against
trajectories produced by the reference's own optimizer.py (tests/golden/adan_*.npz, made by
tests/golden/make_adan_golden.py)."""
import math

import numpy as np

f32 = np.float32


class AdanOracle:
    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=0.0, no_prox=False):
        self.p = [np.array(p, dtype=f32) for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.wd, self.max_grad_norm, self.no_prox = weight_decay, max_grad_norm, no_prox
        self.m = [np.zeros_like(p) for p in self.p]
        self.n = [np.zeros_like(p) for p in self.p]
        self.d = [np.zeros_like(p) for p in self.p]
        self.prev = [None] * len(self.p)
        self.t = 0

    def step(self, grads):
        grads = [np.array(g, dtype=f32) for g in grads]
        clip = f32(1.0)
        if self.max_grad_norm > 0:  # optimizer.py:133-150
            tot = f32(0)
            for g in grads:
                tot = f32(tot + f32((g * g).sum(dtype=f32)))
            clip = min(f32(self.max_grad_norm) / f32(np.sqrt(tot) + f32(self.eps)), f32(1.0))
        self.t += 1
        b1, b2, b3 = self.betas
        bc1, bc2, bc3 = 1 - b1 ** self.t, 1 - b2 ** self.t, 1 - b3 ** self.t
        bc3s = f32(math.sqrt(bc3))
        step, step_d = f32(-(self.lr / bc1)), f32(-(self.lr * b2 / bc2))
        for i, g in enumerate(grads):
            g = g * f32(clip)
            if self.prev[i] is None or self.t == 1:
                self.prev[i] = -g
            diff = self.prev[i] + g
            self.m[i] = self.m[i] * f32(b1) + f32(1 - b1) * g
            self.d[i] = self.d[i] * f32(b2) + f32(1 - b2) * diff
            u = diff * f32(b2) + g
            self.n[i] = self.n[i] * f32(b3) + f32(1 - b3) * (u * u)
            denom = np.sqrt(self.n[i]) / bc3s + f32(self.eps)
            p = self.p[i]
            if self.no_prox:
                p = p * f32(1 - self.lr * self.wd)
            p = p + step * (self.m[i] / denom)
            p = p + step_d * (self.d[i] / denom)
            if not self.no_prox:
                p = p / f32(1 + self.lr * self.wd)
            self.p[i] = p.astype(f32)
            self.prev[i] = -g
