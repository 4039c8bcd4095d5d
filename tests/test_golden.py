"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
CPU oracle; parity with the reference is unpinned -- it ships no vectors for this path).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the
operator boundary, with the caller's own sequence around it."""
import glob
import math
import os

import numpy as np
import pytest
import torch

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith(("adan_", "caller_")))
IDS = [os.path.basename(p)[:-4] for p in GOLDEN]


def _inputs(kind, d):
    bound = torch.tensor([0.5, 0.0, 0.5])
    p = [torch.from_numpy(d[f"param{i}"]) for i in range(3 if kind == "cholesky" else 4)]
    if kind == "cholesky":
        return p, (torch.tanh(p[0]), p[1] + bound), p[2]
    return p, (torch.tanh(p[0]), torch.abs(p[1] + 0.5), torch.sigmoid(p[2]) * 2 * math.pi), p[3]


def test_fixture_set_is_complete():
    assert len(GOLDEN) == 8


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_oracle_reproduces_golden(oracle, path):
    d = np.load(path)
    kind = os.path.basename(path).split("_")[0]
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    _, op_in, col = _inputs(kind, d)
    proj = (oracle.project_cholesky_fwd if kind == "cholesky" else oracle.project_rs_fwd)(
        *[t.numpy() for t in op_in], H, W)
    assert np.array_equal(proj[2], d["radii"]) and np.array_equal(proj[4], d["num_tiles_hit"])
    np.testing.assert_allclose(proj[0], d["xys"], atol=1e-5)
    np.testing.assert_allclose(proj[3], d["conics"], rtol=1e-5, atol=1e-7)
    offs, ids = oracle.bin_tiles(d["xys"], d["radii"], H, W)
    assert np.array_equal(offs, d["tile_offsets"]) and np.array_equal(ids, d["ids"])
    img = oracle.rasterize_sum_fwd(d["xys"], d["conics"], col.numpy(), np.ones((N, 1)), H, W,
                                   offs, ids)
    ok = d["margin"] > 1e-4
    assert np.abs(img - d["image"])[ok].max() <= 1e-6
    # fp32 C restatement against the fp64 differentiable restatement's render
    ref = np.clip(d["image"], 0, 1).transpose(2, 0, 1)[None]
    assert np.abs(ref - d["render64"])[:, :, ok].max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_reproduces_golden(path):
    from gsplat.project_gaussians_2d import project_gaussians_2d
    from gsplat.project_gaussians_2d_scale_rot import project_gaussians_2d_scale_rot
    from gsplat.rasterize_sum import rasterize_gaussians_sum
    dev = "cuda:0"
    d = np.load(path)
    kind = os.path.basename(path).split("_")[0]
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    nparam = 3 if kind == "cholesky" else 4
    params = [torch.from_numpy(d[f"param{i}"]).to(dev).requires_grad_(True)
              for i in range(nparam)]
    if kind == "cholesky":
        bound = torch.tensor([0.5, 0.0, 0.5], device=dev)
        proj = project_gaussians_2d(torch.tanh(params[0]), params[1] + bound, H, W, tb)
    else:
        proj = project_gaussians_2d_scale_rot(torch.tanh(params[0]), torch.abs(params[1] + 0.5),
                                              torch.sigmoid(params[2]) * 2 * math.pi, H, W, tb)
    xys, depths, radii, conics, nth = proj
    np.testing.assert_allclose(xys.detach().cpu().numpy(), d["xys"], atol=2e-4)
    img = rasterize_gaussians_sum(xys, depths, radii, conics, nth, params[-1],
                                  torch.ones(N, 1, device=dev), H, W, 16, 16,
                                  background=torch.ones(3, device=dev), return_alpha=False)
    out = torch.clamp(img, 0, 1).view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()
    loss = torch.nn.functional.mse_loss(out, torch.from_numpy(d["gt"]).to(dev))
    loss.backward()
    assert abs(loss.item() - float(d["loss64"])) <= 2e-6
    for i, p in enumerate(params):
        g, r = p.grad.cpu().numpy(), d[f"grad{i}"]
        assert np.abs(g - r).max() <= 5e-4 * np.abs(r).max(), f"grad{i}"
        # element-wise as well (fp32 chain against the fp64 restatement)
        allow = 1e-3 * np.abs(r) + 2e-5 * np.abs(r).max()
        assert (np.abs(g - r) <= allow).all(), f"grad{i} element-wise"


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_reproduces_golden_operator_outputs_bit_for_bit(path):
    """The activated operator inputs stored in the fixture go to the projection as they are:
    xys, conics, radii, num_tiles_hit and the tile lists must equal the fixture's bit for bit,
    for the Cholesky and the rotation-scale form alike, and the image to 1e-5."""
    from gaussianimage_amd import ops
    dev = "cuda:0"
    d = np.load(path)
    kind = os.path.basename(path).split("_")[0]
    H, W, N = int(d["H"]), int(d["W"]), int(d["N"])
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    ins = [torch.from_numpy(d[f"in{i}"]).to(dev) for i in range(2 if kind == "cholesky" else 3)]
    proj = (ops.project_gaussians_2d if kind == "cholesky"
            else ops.project_gaussians_2d_scale_rot)(*ins, H, W, tb)
    xys, depths, radii, conics, nth = proj
    for got, name in ((xys, "xys"), (radii, "radii"), (conics, "conics"),
                      (nth, "num_tiles_hit")):
        assert np.array_equal(got.cpu().numpy(), d[name]), name
    col = torch.from_numpy(d[f"param{2 if kind == 'cholesky' else 3}"]).to(dev)
    img = ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, col,
                                      torch.ones(N, 1, device=dev), H, W, 16, 16)
    offs, ids = ops.get_workspace(dev).bins()
    assert np.array_equal(offs.numpy(), d["tile_offsets"]) and np.array_equal(ids.numpy(), d["ids"])
    ok = d["margin"] > 1e-4
    assert np.abs(img.cpu().numpy() - d["image"])[ok].max() <= 1e-5
