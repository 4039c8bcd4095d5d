"""Adan: golden trajectories made by the reference's own optimizer.py
(tests/golden/make_adan_golden.py).  CPU: the numpy restatement reproduces them (this pins the
restatement).  GPU: the fused HIP step reproduces them through gaussianimage_amd.Adan."""
import glob
import os

import numpy as np
import pytest
import torch

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "adan_*.npz")))
IDS = [os.path.basename(p)[5:-4] for p in GOLDEN]
RTOL = 2e-5  # fp32 with/without fused multiply-adds over 8 steps


def _kw(d):
    kw = {k: float(v) for k, v in zip(d["kw_keys"], d["kw_vals"])}
    if "no_prox" in kw:
        kw["no_prox"] = bool(kw["no_prox"])
    return kw


def _close(a, b, what):
    scale = max(float(np.abs(b).max()), 1e-12)
    assert float(np.abs(a - b).max()) <= RTOL * scale, what


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_numpy_restatement_matches_reference_trajectory(path):
    from oracle.adan_oracle import AdanOracle
    d = np.load(path)
    assert len(GOLDEN) == 4
    nt = 4
    opt = AdanOracle([d[f"p{i}_init"] for i in range(nt)], lr=float(d["lr"]), **_kw(d))
    for t in range(int(d["steps"])):
        if t == 5:
            opt.lr *= 0.5
        opt.step([d[f"g{i}_t{t}"] for i in range(nt)])
        for i in range(nt):
            _close(opt.p[i], d[f"p{i}_t{t}"], f"param {i} after step {t}")
    for i in range(nt):
        _close(opt.m[i], d[f"exp_avg{i}"], "exp_avg")
        _close(opt.n[i], d[f"exp_avg_sq{i}"], "exp_avg_sq")
        _close(opt.d[i], d[f"exp_avg_diff{i}"], "exp_avg_diff")
        _close(opt.prev[i], d[f"neg_pre_grad{i}"], "neg_pre_grad")


def test_constructor_checks_match_reference():
    from gaussianimage_amd.adan import Adan
    p = [torch.nn.Parameter(torch.zeros(3))]
    for bad in (dict(lr=-1.0), dict(eps=-1.0), dict(betas=(1.0, 0.9, 0.9)),
                dict(betas=(0.9, 0.9, -0.1)), dict(max_grad_norm=-1.0)):
        with pytest.raises(ValueError):
            Adan(p, **bad)
    opt = Adan(p, lr=1e-3)
    p[0].grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_fused_hip_step_matches_reference_trajectory(path):
    from gaussianimage_amd.adan import Adan
    dev = "cuda:0"
    d = np.load(path)
    nt = 4
    params = [torch.nn.Parameter(torch.from_numpy(d[f"p{i}_init"]).to(dev)) for i in range(nt)]
    opt = Adan(params, lr=float(d["lr"]), **_kw(d))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.5)
    for t in range(int(d["steps"])):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(d[f"g{i}_t{t}"]).to(dev)
        opt.step()
        sched.step()
        for i, p in enumerate(params):
            _close(p.detach().cpu().numpy(), d[f"p{i}_t{t}"], f"param {i} after step {t}")
    for i, p in enumerate(params):
        st = opt.state[p]
        for k in ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"):
            _close(st[k].cpu().numpy(), d[f"{k}{i}"], k)
