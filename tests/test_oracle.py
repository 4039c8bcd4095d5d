"""CPU checks of the oracle itself: the C restatement against the differentiable
PyTorch restatement (values, and analytic backward against autograd in fp64),
binning invariants, and the edge cases of the domain (empty input, Gaussians
off the image, zero determinant, ragged edge tiles)."""
import numpy as np
import pytest
import torch

from gaussianimage_amd._testing import synth
from oracle import torch_oracle as to

SIZES = [(48, 64), (80, 96), (33, 50)]


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-30))


@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("seed", [0, 1])
def test_project_cholesky_matches_torch(oracle, hw, seed):
    H, W = hw
    means, L, _, _ = synth.cholesky_inputs(200, seed=seed, footprint=1.0 + seed)
    xys, dep, rad, con, nth = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    t = to.project_gaussians_2d(means, L, H, W, oracle.tile_bounds(H, W))
    assert np.array_equal(rad, t[2].numpy())
    assert np.array_equal(nth, t[4].numpy())
    np.testing.assert_allclose(xys, t[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(con, t[3].numpy(), rtol=1e-5, atol=1e-6)
    assert not dep.any()


@pytest.mark.parametrize("hw", SIZES)
def test_project_rs_matches_torch(oracle, hw):
    H, W = hw
    means, sc, rot, _, _ = synth.rs_inputs(200, seed=3)
    xys, dep, rad, con, nth = oracle.project_rs_fwd(means.numpy(), sc.numpy(), rot.numpy(), H, W)
    t = to.project_gaussians_2d_scale_rot(means, sc, rot, H, W, oracle.tile_bounds(H, W))
    # sin/cos differ in the last bit between libm and torch: allow rare +-1 on the ceil
    assert (rad != t[2].numpy()).sum() <= 1
    np.testing.assert_allclose(xys, t[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(con, t[3].numpy(), rtol=2e-5, atol=1e-6)


def test_binning_invariants(oracle):
    H, W = 80, 96
    means, L, _, _ = synth.cholesky_inputs(500, seed=5, footprint=2.0)
    xys, _, rad, _, nth = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    offs, ids = oracle.bin_tiles(xys, rad, H, W)
    tb = oracle.tile_bounds(H, W)
    assert offs[0] == 0 and offs[-1] == len(ids) == nth.sum()
    assert np.all(np.diff(offs) >= 0)
    boxes = to.tile_boxes(torch.from_numpy(xys), torch.from_numpy(rad), tb).numpy()
    for t in range(tb[0] * tb[1]):
        lst = ids[offs[t]:offs[t + 1]]
        assert np.all(np.diff(lst) > 0), "ascending Gaussian id inside a tile"
        tx, ty = t % tb[0], t // tb[0]
        for g in lst:
            assert boxes[g, 0] <= tx < boxes[g, 1] and boxes[g, 2] <= ty < boxes[g, 3]
    # every Gaussian appears num_tiles_hit times
    assert np.array_equal(np.bincount(ids, minlength=500), nth)


@pytest.mark.parametrize("hw", SIZES)
def test_raster_forward_matches_torch(oracle, hw):
    H, W = hw
    means, L, col, op = synth.cholesky_inputs(150, seed=2, footprint=1.5)
    op = op * torch.rand(150, 1, generator=torch.Generator().manual_seed(0))
    xys, _, rad, con, _ = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    offs, ids = oracle.bin_tiles(xys, rad, H, W)
    out, margin = oracle.rasterize_sum_fwd(xys, con, col.numpy(), op.numpy(), H, W, offs, ids,
                                           want_margin=True)
    ref = to.rasterize_gaussians_sum(torch.from_numpy(xys).double(), torch.from_numpy(rad),
                                     torch.from_numpy(con).double(), col.double(), op.double(), H,
                                     W).numpy()
    ok = margin > 1e-4  # pixels whose value does not hinge on the last bits of exp()
    assert ok.mean() > 0.99
    assert np.abs(out - ref)[ok].max() < 1e-5


def test_raster_backward_matches_autograd(oracle):
    H, W, N = 48, 64, 96
    means, L, col, op = synth.cholesky_inputs(N, seed=4, footprint=1.5)
    op = 0.3 + 0.7 * torch.rand(N, 1, generator=torch.Generator().manual_seed(1))
    xys, _, rad, con, _ = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    offs, ids = oracle.bin_tiles(xys, rad, H, W)
    v_out = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(2))
    leaves = [torch.from_numpy(xys).double().requires_grad_(True),
              torch.from_numpy(con).double().requires_grad_(True),
              col.double().requires_grad_(True), op.double().requires_grad_(True)]
    img = to.rasterize_gaussians_sum(leaves[0], torch.from_numpy(rad), leaves[1], leaves[2],
                                     leaves[3], H, W)
    (img * v_out.double()).sum().backward()
    got = oracle.rasterize_sum_bwd(xys, con, col.numpy(), op.numpy(), H, W, offs, ids,
                                   v_out.numpy())
    for g, leaf in zip(got, leaves):
        assert _rel(g, leaf.grad.numpy()) < 1e-5


def test_project_cholesky_backward_matches_autograd(oracle):
    H, W, N = 48, 64, 64
    means, L, _, _ = synth.cholesky_inputs(N, seed=6)
    gen = torch.Generator().manual_seed(3)
    v_xy, v_con = torch.randn(N, 2, generator=gen), torch.randn(N, 3, generator=gen)
    m, l = means.double().requires_grad_(True), L.double().requires_grad_(True)
    xys, _, _, con, _ = to.project_gaussians_2d(m, l, H, W, oracle.tile_bounds(H, W))
    ((xys * v_xy).sum() + (con * v_con).sum()).backward()
    v_means, v_L = oracle.project_cholesky_bwd(means.numpy(), L.numpy(), H, W, v_xy.numpy(),
                                               v_con.numpy())
    assert _rel(v_means, m.grad.numpy()) < 1e-5
    assert _rel(v_L, l.grad.numpy()) < 1e-5


def test_project_rs_backward_matches_autograd(oracle):
    H, W, N = 48, 64, 64
    means, sc, rot, _, _ = synth.rs_inputs(N, seed=7)
    gen = torch.Generator().manual_seed(4)
    v_xy, v_con = torch.randn(N, 2, generator=gen), torch.randn(N, 3, generator=gen)
    m, s, r = (means.double().requires_grad_(True), sc.double().requires_grad_(True),
               rot.double().requires_grad_(True))
    xys, _, _, con, _ = to.project_gaussians_2d_scale_rot(m, s, r, H, W, oracle.tile_bounds(H, W))
    ((xys * v_xy).sum() + (con * v_con).sum()).backward()
    v_means, v_s, v_r = oracle.project_rs_bwd(means.numpy(), sc.numpy(), rot.numpy(), H, W,
                                              v_xy.numpy(), v_con.numpy())
    assert _rel(v_means, m.grad.numpy()) < 1e-5
    assert _rel(v_s, s.grad.numpy()) < 2e-5
    assert _rel(v_r, r.grad.numpy()) < 2e-5


def test_gradcheck_fp64_end_to_end():
    """The differentiable restatement itself against finite differences."""
    H, W, N = 32, 32, 6
    xyz, chol, feat = synth.cholesky_params(N, seed=8)
    args = [t.double().requires_grad_(True) for t in (xyz, chol, feat)]
    conv = dict(to.DEFAULT_CONV, alpha_min=0.0)  # no hard threshold inside the FD stencil

    def f(a, b, c):
        tbnd = ((W + 15) // 16, (H + 15) // 16, 1)
        xys, _, radii, con, _ = to.project_gaussians_2d(torch.tanh(a), b + torch.tensor(
            [0.5, 0.0, 0.5], dtype=torch.float64), H, W, tbnd, conv)
        radii = torch.full_like(radii, 64)  # every tile: keeps membership fixed under FD
        return to.rasterize_gaussians_sum(xys, radii, con, c, torch.ones(N, 1, dtype=torch.float64),
                                          H, W, None, conv)

    assert torch.autograd.gradcheck(f, args, eps=1e-6, atol=1e-5, rtol=1e-4)


def test_edge_cases(oracle):
    H, W = 40, 56
    # empty input
    xys, dep, rad, con, nth = oracle.project_cholesky_fwd(np.zeros((0, 2)), np.zeros((0, 3)), H, W)
    offs, ids = oracle.bin_tiles(xys, rad, H, W)
    assert len(ids) == 0 and not offs.any()
    out = oracle.rasterize_sum_fwd(xys, con, np.zeros((0, 3)), np.zeros((0, 1)), H, W, offs, ids)
    assert out.shape == (H, W, 3) and not out.any()
    # zero determinant -> early-out leaves zeros; far outside the image -> no tiles
    means = np.array([[0.0, 0.0], [30.0, 30.0], [0.2, -0.3]], np.float32)
    L = np.array([[1.0, 0.5, 0.0], [1.0, 0.0, 1.0], [0.8, 0.1, 0.9]], np.float32)
    xys, dep, rad, con, nth = oracle.project_cholesky_fwd(means, L, H, W)
    assert rad[0] == 0 and nth[0] == 0 and not con[0].any() and not xys[0].any()
    assert rad[1] > 0 and nth[1] == 0
    assert rad[2] > 0 and nth[2] > 0
    offs, ids = oracle.bin_tiles(xys, rad, H, W)
    assert set(ids.tolist()) == {2}
    # NaN parameters are treated as the early-out
    L[2, 0] = np.nan
    _, _, rad, _, nth = oracle.project_cholesky_fwd(means, L, H, W)
    assert rad[2] == 0 and nth[2] == 0


def test_conventions_are_runtime(oracle):
    """pixel_offset and center_bias form a coupled pair: shifting both by +0.5/-0.5... no:
    shifting the sample point by d and the centres by d leaves the image unchanged where the
    tile lists do not change."""
    H, W = 32, 32
    means, L, col, op = synth.cholesky_inputs(40, seed=9)
    means = means * 0.5  # keep everything inside one tile ring so lists do not change
    c0 = oracle.default_conv()
    c1 = oracle.default_conv(pixel_offset=0.25, center_bias=0.25)
    outs = []
    for c in (c0, c1):
        xys, _, rad, con, _ = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W, c)
        rad[:] = 64
        offs, ids = oracle.bin_tiles(xys, rad, H, W)
        outs.append(oracle.rasterize_sum_fwd(xys, con, col.numpy(), op.numpy(), H, W, offs, ids,
                                             conv=c))
    assert np.abs(outs[0] - outs[1]).max() < 1e-5


def test_shared_sincos_against_libm(oracle):
    """The fixed-sequence sine / cosine that the kernels and the C restatement share: within
    1.5e-7 of the double-precision functions for |theta| <= 1e5 (dense near [0, 2 pi], the
    model's range: sigmoid * 2 pi, gaussianimage_rs.py:59-60), NaN outside +-1e9 and for
    non-finite angles, and exact at zero."""
    rng = np.random.default_rng(0)
    theta = np.concatenate([np.linspace(-7, 7, 400001), rng.uniform(-1e5, 1e5, 400000),
                            rng.uniform(-300, 300, 200000)]).astype(np.float32)
    s, c = oracle.sincos_f32(theta)
    t64 = theta.astype(np.float64)
    assert np.abs(s - np.sin(t64)).max() <= 1.5e-7
    assert np.abs(c - np.cos(t64)).max() <= 1.5e-7
    s, c = oracle.sincos_f32(np.array([0.0, 2e9, -2e9, np.inf, -np.inf, np.nan], np.float32))
    assert s[0] == 0.0 and c[0] == 1.0
    assert np.isnan(s[1:]).all() and np.isnan(c[1:]).all()
    # accuracy degrades gracefully up to the 1e9 limit (double-precision reduction)
    big = rng.uniform(-1e9, 1e9, 100000).astype(np.float32)
    s, c = oracle.sincos_f32(big)
    assert np.abs(s - np.sin(big.astype(np.float64))).max() <= 1e-6
