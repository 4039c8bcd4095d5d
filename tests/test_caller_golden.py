"""The caller's sequence around the operator boundary, pinned by fixtures that the REFERENCE'S OWN
model files produced (tests/golden/caller_*.npz, made by tests/golden/make_caller_golden.py: it
imports /root/reference/gaussianimage_cholesky.py and gaussianimage_rs.py unchanged and runs
forward() and three train_iter() with the reference's Adan and StepLR; the three gsplat
operators, whose source the reference does not contain, resolve there to this repository's CPU
restatement in float64 -- the operators stay parity-unpinned, the sequence around them is the
reference's).

CPU: the C restatement reproduces the fixtures' first forward and first gradients.
GPU: the model stand-ins (gaussianimage_amd/_testing/harness.py) through the drop-in operators,
and the fused fitting step, reproduce images, losses, gradients and updated parameters."""
import glob
import math
import os

import numpy as np
import pytest
import torch

CALLER = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "caller_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in CALLER]
DEV = "cuda:0"
LR = 1e-3


def _kind(path):
    return os.path.basename(path).split("_")[1]


def _activated(kind, d):
    """Operator inputs from the raw initial parameters, as the reference's properties compute
    them (gaussianimage_cholesky.py:62,74; gaussianimage_rs.py:55-60)."""
    xyz = torch.from_numpy(d["init_xyz"])
    if kind == "cholesky":
        return (torch.tanh(xyz), torch.from_numpy(d["init_cholesky"]) +
                torch.tensor([0.5, 0.0, 0.5]))
    return (torch.tanh(xyz), torch.abs(torch.from_numpy(d["init_scaling"]) + 0.5),
            torch.sigmoid(torch.from_numpy(d["init_rotation"])) * 2 * math.pi)


def _oracle_forward(oracle, kind, d, colors):
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    ins = [t.numpy() for t in _activated(kind, d)]
    proj = (oracle.project_cholesky_fwd if kind == "cholesky" else oracle.project_rs_fwd)(
        *ins, H, W)
    offs, ids = oracle.bin_tiles(proj[0], proj[2], H, W)
    img, margin = oracle.rasterize_sum_fwd(proj[0], proj[3], colors, np.ones((N, 1), np.float32),
                                           H, W, offs, ids, want_margin=True)
    return ins, proj, offs, ids, img, margin


def test_fixture_set_is_complete():
    assert len(CALLER) == 4
    for p in CALLER:
        d = np.load(p)
        assert int(d["iters"]) == 3 and float(d["lr_after"]) == LR   # StepLR: no decay yet
        assert d["loss_3"] < d["loss_1"]


@pytest.mark.parametrize("path", CALLER, ids=IDS)
def test_oracle_reproduces_the_reference_callers_first_forward_and_gradients(oracle, path):
    d = np.load(path)
    kind = _kind(path)
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    col = d["init_features_dc"]
    ins, proj, offs, ids, img, margin = _oracle_forward(oracle, kind, d, col)
    assert np.array_equal(proj[2], d["radii"])
    np.testing.assert_allclose(proj[0], d["xys"], atol=1e-5)
    ok = margin > 1e-4
    assert ok.mean() > 0.99
    render = np.clip(img, 0, 1).transpose(2, 0, 1)[None]
    assert np.abs(render - d["forward_render"])[:, :, ok].max() <= 1e-5
    diff = render - d["gt"]
    loss = float((diff.astype(np.float64) ** 2).mean())
    assert abs(loss - float(d["loss_1"])) <= 2e-6
    # the first iteration's gradients through the C restatement's analytic backward
    inside = ((img > 0) & (img < 1)).astype(np.float32)       # clamp passes gradient inside [0,1]
    v_img = (2.0 * diff[0].transpose(1, 2, 0) / diff.size * inside).astype(np.float32)
    v_xy, v_conic, v_col, _ = oracle.rasterize_sum_bwd(proj[0], proj[3], col,
                                                       np.ones((N, 1), np.float32), H, W, offs,
                                                       ids, v_img)

    def close(got, want, name):
        m = np.abs(want).max()
        assert np.abs(got - want).max() <= 5e-4 * m, name

    close(v_col, d["grad_1_features_dc"], "colours")
    xyz = torch.from_numpy(d["init_xyz"])
    dtanh = (1 - torch.tanh(xyz) ** 2).numpy()
    if kind == "cholesky":
        v_means, v_L = oracle.project_cholesky_bwd(ins[0], ins[1], H, W, v_xy, v_conic)
        close(v_L, d["grad_1_cholesky"], "cholesky")
    else:
        v_means, v_sc, v_rot = oracle.project_rs_bwd(ins[0], ins[1], ins[2], H, W, v_xy, v_conic)
        sign = np.sign(d["init_scaling"] + 0.5)
        close(v_sc * sign, d["grad_1_scaling"], "scaling")
        sg = torch.sigmoid(torch.from_numpy(d["init_rotation"])).numpy()
        close(v_rot.reshape(N, 1) * (2 * math.pi * sg * (1 - sg)), d["grad_1_rotation"],
              "rotation")
    close(v_means * dtanh, d["grad_1_xyz"], "xyz")


def _params_close(got, want, k_iters, name):
    """Two float32/float64 executions of the same Adan trajectory: Adan divides by
    sqrt(n) + 1e-8, so an element whose gradient is rounding noise moves by about lr per step in
    a direction the last bits decide.  All but a few elements agree to 2e-4 of the tensor's
    range, and none is further off than the steps taken allow."""
    a, b = np.asarray(got, np.float64), np.asarray(want, np.float64)
    diff = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
    assert (diff <= 2e-4).mean() >= 0.998, (name, float((diff <= 2e-4).mean()))
    assert np.abs(a - b).max() <= 2.2 * k_iters * LR, name


@pytest.mark.gpu
@pytest.mark.parametrize("path", CALLER, ids=IDS)
def test_model_stand_ins_reproduce_the_reference_callers(oracle, path):
    from gaussianimage_amd._testing import harness
    d = np.load(path)
    kind = _kind(path)
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    names = [str(n) for n in d["names"]]
    model = (harness.CholeskyImageModel if kind == "cholesky" else harness.RSImageModel)(
        N, H, W, DEV, lr=LR)
    with torch.no_grad():
        for k in names:
            getattr(model, k).copy_(torch.from_numpy(d["init" + k]))
    if kind == "cholesky":
        model.random_colors = torch.from_numpy(d["random_colors"])
    gt = torch.from_numpy(d["gt"]).to(DEV)
    model.eval()
    with torch.no_grad():
        pkg = model()
    *_, margin = _oracle_forward(oracle, kind, d, d["init_features_dc"])
    ok = margin > 1e-4
    for k in ("render", "gauss_render", "alpha_map"):
        if "forward_" + k in d.files:
            got = pkg[k].cpu().numpy()
            assert got.shape == d["forward_" + k].shape, k
            assert np.abs(got - d["forward_" + k])[:, :, ok].max() <= 1e-5, k
    np.testing.assert_allclose(model.xys.cpu().numpy(), d["xys"], atol=2e-4)
    assert np.array_equal(model.radii.cpu().numpy(), d["radii"])
    grads = {}
    model.optimizer.register_step_pre_hook(
        lambda opt, a, kw: grads.update({k: getattr(model, k).grad.detach().cpu().numpy().copy()
                                         for k in names}))
    model.train()
    for it in (1, 2, 3):
        loss, psnr = model.train_iter(gt)
        assert abs(loss.item() - float(d[f"loss_{it}"])) <= 2e-6, it
        assert abs(psnr - float(d[f"psnr_{it}"])) <= 1e-3, it
        if it == 1:   # (later iterations start from parameters that already differ by rounding)
            for k in names:
                r = d[f"grad_1{k}"]
                assert np.abs(grads[k] - r).max() <= 5e-4 * np.abs(r).max(), k
        for k in names:
            _params_close(getattr(model, k).detach().cpu().numpy(), d[f"param_{it}{k}"], it, k)
    assert model.optimizer.param_groups[0]["lr"] == float(d["lr_after"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", CALLER, ids=IDS)
def test_fused_fitter_reproduces_the_reference_callers(path):
    from gaussianimage_amd.fit import FusedFitter
    d = np.load(path)
    kind = _kind(path)
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    names = [str(n) for n in d["names"]]
    fit = FusedFitter(N, H, W, kind=kind, device=DEV, lr=LR,
                      params={k: torch.from_numpy(d["init" + k]) for k in names})
    gt = torch.from_numpy(d["gt"]).to(DEV)
    for it in (1, 2, 3):
        fit.step(gt)
        assert abs(fit.loss() - float(d[f"loss_{it}"])) <= 2e-6, it
        assert abs(fit.psnr() - float(d[f"psnr_{it}"])) <= 1e-3, it
        for k in names:
            _params_close(fit.params[k].cpu().numpy(), d[f"param_{it}{k}"], it, k)
    assert fit.lr == float(d["lr_after"])
