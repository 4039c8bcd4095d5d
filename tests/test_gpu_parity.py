"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer outputs (radii, tile counts, tile lists) are compared bit for bit;
the image to 1e-5 absolute and gradients to 1e-4 relative (the tolerances the task states).

Pixels whose value hinges on the last bits of exp() at the alpha >= 1/255 cut (the oracle
reports that margin) are set aside in image comparisons: no two exp implementations agree
there, and a flipped pair moves a pixel by ~1/255."""
import numpy as np
import pytest
import torch

from gaussianimage_amd._testing import synth

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
IMG_ATOL = 1e-5
GRAD_RTOL = 1e-4


def _np(t):
    return t.detach().cpu().numpy()


def _maxrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


GRAD_FLOOR = 2e-6


def _grad_excess(got, ref, rtol=GRAD_RTOL, floor=GRAD_FLOOR):
    """Element-wise gradient criterion: |got - ref| <= rtol * |ref| + floor * max|ref| for every
    element (the floor is what fp32 sums of cancelling terms can resolve: a small element of a
    tensor whose largest is M carries rounding of the order of 1e-7 * M per hundred terms).
    Returns the largest ratio of an element's error to its allowance (<= 1 passes)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    allow = rtol * np.abs(ref) + floor * max(np.abs(ref).max(), 1e-30)
    return float((np.abs(got - ref) / allow).max())


def _assert_grad(got, ref, name=""):
    assert _maxrel(got, ref) <= GRAD_RTOL, name
    assert _grad_excess(got, ref) <= 1.0, name


@pytest.fixture(scope="module")
def ops():
    from gaussianimage_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


CASES = [(48, 64, 64, 0, 1.0), (80, 96, 512, 1, 1.0), (33, 50, 300, 2, 2.0),
         (512, 768, 5000, 1, 1.0), (100, 130, 2000, 3, 4.0)]


@pytest.mark.parametrize("H,W,N,seed,fp", CASES)
def test_project_cholesky_forward(ops, oracle, H, W, N, seed, fp):
    means, L, _, _ = synth.cholesky_inputs(N, seed=seed, footprint=fp)
    tb = oracle.tile_bounds(H, W)
    got = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    ref = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    assert np.array_equal(_np(got[2]), ref[2]), "radii"
    assert np.array_equal(_np(got[4]), ref[4]), "num_tiles_hit"
    assert np.array_equal(_np(got[0]), ref[0]), "xys"
    assert np.array_equal(_np(got[3]), ref[3]), "conics"
    assert not _np(got[1]).any()
    assert got[2].dtype == torch.int32 and got[4].dtype == torch.int32


@pytest.mark.parametrize("H,W,N,seed,fp", CASES)
def test_project_rs_forward(ops, oracle, H, W, N, seed, fp):
    means, sc, rot, _, _ = synth.rs_inputs(N, seed=seed, footprint=fp)
    tb = oracle.tile_bounds(H, W)
    got = ops.project_gaussians_2d_scale_rot(means.to(DEV), sc.to(DEV), rot.to(DEV), H, W, tb)
    ref = oracle.project_rs_fwd(means.numpy(), sc.numpy(), rot.numpy(), H, W)
    # the projection's sine / cosine is one fixed IEEE sequence shared with the oracle: every
    # output is bit-exact, as for the Cholesky form
    assert np.array_equal(_np(got[2]), ref[2]), "radii"
    assert np.array_equal(_np(got[4]), ref[4]), "num_tiles_hit"
    assert np.array_equal(_np(got[0]), ref[0]), "xys"
    assert np.array_equal(_np(got[3]), ref[3]), "conics"
    assert not _np(got[1]).any()


def test_project_rs_forward_wide_angles(ops, oracle):
    """Angles far outside [0, 2 pi), negative, tiny, huge (NaN results -> early-out) and
    non-finite: still bit-exact."""
    N, H, W = 4096, 80, 96
    means, sc, rot, _, _ = synth.rs_inputs(N, seed=9)
    g = torch.Generator().manual_seed(10)
    rot = (torch.rand(N, 1, generator=g) - 0.5) * torch.logspace(-6, 10, N).reshape(N, 1)
    rot[::97] = float("inf")
    rot[1::97] = float("nan")
    tb = oracle.tile_bounds(H, W)
    got = ops.project_gaussians_2d_scale_rot(means.to(DEV), sc.to(DEV), rot.to(DEV), H, W, tb)
    ref = oracle.project_rs_fwd(means.numpy(), sc.numpy(), rot.numpy(), H, W)
    for gt_, r, name in zip(got, ref, ("xys", "depths", "radii", "conics", "num_tiles_hit")):
        assert np.array_equal(_np(gt_), r), name
    assert (ref[2][::97] == 0).all() and (ref[2] > 0).sum() > N // 2


def _project_and_bin(ops, oracle, H, W, N, seed, fp):
    means, L, col, op = synth.cholesky_inputs(N, seed=seed, footprint=fp)
    tb = oracle.tile_bounds(H, W)
    xys, depths, radii, conics, nth = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    ws = ops.get_workspace(DEV)
    ws.bin(xys, radii, H, W, force=True)
    return means, L, col, op, (xys, depths, radii, conics, nth), ws


@pytest.mark.parametrize("H,W,N,seed,fp", CASES + [(64, 64, 3000, 5, 6.0)])
def test_tile_lists_match_bit_for_bit(ops, oracle, H, W, N, seed, fp):
    *_, proj, ws = _project_and_bin(ops, oracle, H, W, N, seed, fp)
    offs, ids = ws.bins()
    r_offs, r_ids = oracle.bin_tiles(_np(proj[0]), _np(proj[2]), H, W)
    assert np.array_equal(offs.numpy(), r_offs)
    assert np.array_equal(ids.numpy(), r_ids)
    assert int(offs[-1]) == len(r_ids) == int(_np(proj[4]).sum())


def test_tile_lists_long_lists_and_standalone_count(ops, oracle):
    """Lists longer than one wave (LDS sort) and longer than the LDS sort limit (in-place sort
    in memory); binning from caller-made tensors (no fused count from a projection)."""
    H, W, N = 48, 48, 6000
    g = torch.Generator().manual_seed(11)
    xys = torch.rand(N, 2, generator=g) * torch.tensor([W, H])
    radii = torch.randint(1, 40, (N,), generator=g, dtype=torch.int32)
    radii[::7] = 0
    ws = ops.get_workspace(DEV)
    ws.bin(xys.to(DEV), radii.to(DEV), H, W, force=True)
    offs, ids = ws.bins()
    r_offs, r_ids = oracle.bin_tiles(xys.numpy(), radii.numpy(), H, W)
    assert np.diff(r_offs).max() > 4096 and np.diff(r_offs).min() > 64
    assert np.array_equal(offs.numpy(), r_offs)
    assert np.array_equal(ids.numpy(), r_ids)


@pytest.mark.parametrize("H,W,N,seed,fp", CASES)
def test_rasterize_forward(ops, oracle, H, W, N, seed, fp):
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, seed, fp)
    op = 0.2 + 0.8 * torch.rand(N, 1, generator=torch.Generator().manual_seed(seed))
    xys, depths, radii, conics, nth = proj
    img = ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, col.to(DEV), op.to(DEV), H,
                                      W, 16, 16, background=torch.ones(3, device=DEV))
    assert img.shape == (H, W, 3) and img.dtype == torch.float32 and img.is_contiguous()
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    ref, margin = oracle.rasterize_sum_fwd(_np(xys), _np(conics), col.numpy(), op.numpy(), H, W,
                                           offs, ids, want_margin=True)
    ok = margin > 1e-4
    assert ok.mean() > 0.995
    err = np.abs(_np(img) - ref)
    assert err[ok].max() <= IMG_ATOL
    # a set-aside pixel can differ by at most a few cut-off contributions
    assert err.max() <= 3.0 * (1 / 255) * max(1.0, float(col.max()))


@pytest.mark.parametrize("H,W,N,seed,fp", CASES)
def test_rasterize_backward(ops, oracle, H, W, N, seed, fp):
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, seed, fp)
    op = 0.2 + 0.8 * torch.rand(N, 1, generator=torch.Generator().manual_seed(seed))
    xys, depths, radii, conics, nth = proj
    v_out = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(seed + 100))
    leaves = [xys.detach().clone().requires_grad_(True),
              conics.detach().clone().requires_grad_(True),
              col.to(DEV).requires_grad_(True), op.to(DEV).requires_grad_(True)]
    img = ops.rasterize_gaussians_sum(leaves[0], depths, radii, leaves[1], nth, leaves[2],
                                      leaves[3], H, W, 16, 16)
    (img * v_out.to(DEV)).sum().backward()
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    ref = oracle.rasterize_sum_bwd(_np(xys), _np(conics), col.numpy(), op.numpy(), H, W, offs, ids,
                                   v_out.numpy())
    for name, leaf, r in zip(("xys", "conics", "colors", "opacity"), leaves, ref):
        assert leaf.grad.shape == leaf.shape
        _assert_grad(_np(leaf.grad), r.reshape(leaf.shape), name)


def test_backward_culling_changes_nothing(ops, oracle):
    """The support-rectangle culling (forward and backward) drops no contributing pixel."""
    H, W, N = 80, 96, 800
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, 4, 2.5)
    op = (0.05 + 0.95 * torch.rand(N, 1, generator=torch.Generator().manual_seed(0))).to(DEV)
    op[::9] = 0.003  # below the cut: contributes nowhere
    xys, depths, radii, conics, nth = proj
    v_out = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    grads = []
    for cull in (True, False):
        ws.set_cull(cull)
        leaves = [xys.detach().clone().requires_grad_(True),
                  conics.detach().clone().requires_grad_(True),
                  col.to(DEV).requires_grad_(True), op.clone().requires_grad_(True)]
        img = ops.rasterize_gaussians_sum(leaves[0], depths, radii, leaves[1], nth, leaves[2],
                                          leaves[3], H, W, 16, 16)
        (img * v_out).sum().backward()
        grads.append([_np(l.grad) for l in leaves] + [_np(img)])
    ws.set_cull(True)
    # forward: culled entries contribute exactly zero, the order of the rest is unchanged
    assert np.array_equal(grads[0].pop(), grads[1].pop())
    # the culled walk visits the same contributing pixels in another order: equal up to fp32
    # re-association, far inside the gradient tolerance
    for a, b in zip(*grads):
        assert _maxrel(a, b) <= 2e-6


@pytest.mark.parametrize("N,seed", [(64, 0), (3000, 1)])
def test_project_backward(ops, oracle, N, seed):
    H, W = 80, 96
    tb = oracle.tile_bounds(H, W)
    gen = torch.Generator().manual_seed(seed)
    v_xy, v_con = torch.randn(N, 2, generator=gen), torch.randn(N, 3, generator=gen)
    means, L, _, _ = synth.cholesky_inputs(N, seed=seed)
    m, l = means.to(DEV).requires_grad_(True), L.to(DEV).requires_grad_(True)
    xys, _, _, con, _ = ops.project_gaussians_2d(m, l, H, W, tb)
    ((xys * v_xy.to(DEV)).sum() + (con * v_con.to(DEV)).sum()).backward()
    r_m, r_l = oracle.project_cholesky_bwd(means.numpy(), L.numpy(), H, W, v_xy.numpy(),
                                           v_con.numpy())
    _assert_grad(_np(m.grad), r_m, "means")
    _assert_grad(_np(l.grad), r_l, "L")
    means, sc, rot, _, _ = synth.rs_inputs(N, seed=seed)
    m, s, r = (means.to(DEV).requires_grad_(True), sc.to(DEV).requires_grad_(True),
               rot.to(DEV).requires_grad_(True))
    xys, _, _, con, _ = ops.project_gaussians_2d_scale_rot(m, s, r, H, W, tb)
    ((xys * v_xy.to(DEV)).sum() + (con * v_con.to(DEV)).sum()).backward()
    r_m, r_s, r_r = oracle.project_rs_bwd(means.numpy(), sc.numpy(), rot.numpy(), H, W,
                                          v_xy.numpy(), v_con.numpy())
    _assert_grad(_np(m.grad), r_m, "means")
    _assert_grad(_np(s.grad), r_s, "scales")
    _assert_grad(_np(r.grad), r_r, "rotation")


def test_model_sequence_cholesky_end_to_end(ops, oracle):
    """The caller's own sequence (gaussianimage_cholesky.py:62,74,76-85,104,116): tanh / bound
    -> project -> rasterize -> clamp -> permute -> mse -> backward, against the oracle chain."""
    from oracle import torch_oracle as to
    H, W, N = 80, 96, 512
    xyz, chol, feat = synth.cholesky_params(N, seed=2)
    gt = synth.target_image(H, W)
    params = [p.to(DEV).requires_grad_(True) for p in (xyz, chol, feat)]
    tb = oracle.tile_bounds(H, W)
    bound = torch.tensor([0.5, 0.0, 0.5], device=DEV)
    xys, depths, radii, conics, nth = ops.project_gaussians_2d(torch.tanh(params[0]),
                                                               params[1] + bound, H, W, tb)
    out = ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, params[2],
                                      torch.ones(N, 1, device=DEV), H, W, 16, 16,
                                      background=torch.ones(3, device=DEV), return_alpha=False)
    out = torch.clamp(out, 0, 1).view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()
    loss = torch.nn.functional.mse_loss(out, gt.to(DEV))
    loss.backward()
    ref_params = [p.double().requires_grad_(True) for p in (xyz, chol, feat)]
    ref_img = to.model_forward_cholesky(*ref_params, H, W)
    ref_loss = torch.nn.functional.mse_loss(ref_img, gt.double())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-6
    for p, r in zip(params, ref_params):
        assert _maxrel(_np(p.grad), r.grad.numpy()) <= 5e-4


def test_model_sequence_rs_end_to_end(ops, oracle):
    """gaussianimage_rs.py:55-60,74-80,86-87."""
    import math
    from oracle import torch_oracle as to
    H, W, N = 80, 96, 512
    xyz, scaling, rotation, feat = synth.rs_params(N, seed=3)
    gt = synth.target_image(H, W)
    params = [p.to(DEV).requires_grad_(True) for p in (xyz, scaling, rotation, feat)]
    tb = oracle.tile_bounds(H, W)
    xys, depths, radii, conics, nth = ops.project_gaussians_2d_scale_rot(
        torch.tanh(params[0]), torch.abs(params[1] + 0.5), torch.sigmoid(params[2]) * 2 * math.pi,
        H, W, tb)
    out = ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, params[3],
                                      torch.ones(N, 1, device=DEV), H, W, 16, 16,
                                      background=torch.ones(3, device=DEV), return_alpha=False)
    out = torch.clamp(out, 0, 1).view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()
    loss = torch.nn.functional.mse_loss(out, gt.to(DEV))
    loss.backward()
    ref_params = [p.double().requires_grad_(True) for p in (xyz, scaling, rotation, feat)]
    ref_loss = torch.nn.functional.mse_loss(to.model_forward_rs(*ref_params, H, W), gt.double())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 2e-6
    for p, r in zip(params, ref_params):
        assert _maxrel(_np(p.grad), r.grad.numpy()) <= 5e-4


def test_three_passes_share_one_binning_and_runs_are_bit_reproducible(ops, oracle):
    H, W, N = 96, 128, 2000
    means, L, col, op = synth.cholesky_inputs(N, seed=6, footprint=2.0)
    tb = oracle.tile_bounds(H, W)
    imgs, gens = [], []
    for rep in range(2):
        proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
        ws = ops.get_workspace(DEV)
        g0 = ws.info()["generation"]
        a = ops.rasterize_gaussians_sum(*proj, col.to(DEV), op.to(DEV), H, W, 16, 16)
        b = ops.rasterize_gaussians_sum(*proj, col.to(DEV) * 0.5, op.to(DEV), H, W, 16, 16)
        c = ops.rasterize_gaussians_sum(*proj, torch.ones(N, 3, device=DEV), op.to(DEV), H, W, 16,
                                        16)
        gens.append(ws.info()["generation"] - g0)
        imgs.append([_np(a), _np(b), _np(c)])
    assert gens == [1, 1], "one binning per projected frame, reused by the three passes"
    for x, y in zip(*imgs):
        assert np.array_equal(x, y)


def test_empty_and_degenerate_inputs(ops, oracle):
    H, W = 40, 56
    tb = oracle.tile_bounds(H, W)
    z = lambda *s, **k: torch.zeros(*s, device=DEV, **k)
    proj = ops.project_gaussians_2d(z(0, 2), z(0, 3), H, W, tb)
    assert [t.shape[0] for t in proj] == [0] * 5
    img = ops.rasterize_gaussians_sum(*proj, z(0, 3), z(0, 1), H, W, 16, 16)
    assert img.shape == (H, W, 3) and not _np(img).any()
    # zero determinant, off-image, NaN parameter: early-outs with zero outputs
    means = torch.tensor([[0.0, 0.0], [30.0, 30.0], [0.2, -0.3], [0.1, 0.1]])
    L = torch.tensor([[1.0, 0.5, 0.0], [1.0, 0.0, 1.0], [0.8, 0.1, 0.9], [float("nan"), 0.0, 1.0]])
    got = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    ref = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    for g, r in zip(got, ref):
        assert np.array_equal(_np(g), r)
    col = torch.rand(4, 3)
    img = ops.rasterize_gaussians_sum(*got, col.to(DEV), torch.ones(4, 1, device=DEV), H, W, 16,
                                      16)
    offs, ids = oracle.bin_tiles(ref[0], ref[2], H, W)
    rimg = oracle.rasterize_sum_fwd(ref[0], ref[3], col.numpy(), np.ones((4, 1)), H, W, offs, ids)
    assert np.abs(_np(img) - rimg).max() <= IMG_ATOL


def test_fixed_capacity_overflow_is_reported(ops, oracle):
    from gaussianimage_amd import _lib
    H, W, N = 64, 64, 1000
    means, L, col, op = synth.cholesky_inputs(N, seed=8, footprint=3.0)
    tb = oracle.tile_bounds(H, W)
    ws = ops.Workspace(DEV)
    ws.reserve(100, 16)
    proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    with pytest.raises(_lib.Gi2dError) as ei:
        ws.bin(proj[0], proj[2], H, W, flags=_lib.BIN_SYNC | _lib.BIN_NOGROW, force=True)
    assert ei.value.status == _lib.GI2D_E_CAPACITY
    # without NOGROW the same workspace grows and bins correctly
    ws.bin(proj[0], proj[2], H, W, flags=_lib.BIN_SYNC, force=True)
    offs, ids = ws.bins()
    r_offs, r_ids = oracle.bin_tiles(_np(proj[0]), _np(proj[2]), H, W)
    assert np.array_equal(ids.numpy(), r_ids) and np.array_equal(offs.numpy(), r_offs)


def test_baseline_config_full_size(ops, oracle):
    """BASELINE config 2: 768x512, 30k Gaussians, Cholesky, forward + backward."""
    H, W, N = 512, 768, 30000
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, 1, 1.0)
    xys, depths, radii, conics, nth = proj
    leaves = [xys.detach().clone().requires_grad_(True),
              conics.detach().clone().requires_grad_(True), col.to(DEV).requires_grad_(True)]
    opa = torch.ones(N, 1, device=DEV)
    img = ops.rasterize_gaussians_sum(leaves[0], depths, radii, leaves[1], nth, leaves[2], opa, H,
                                      W, 16, 16)
    v_out = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(5))
    (img * v_out.to(DEV)).sum().backward()
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    g_offs, g_ids = ws.bins()
    assert np.array_equal(g_offs.numpy(), offs) and np.array_equal(g_ids.numpy(), ids)
    ref, margin = oracle.rasterize_sum_fwd(_np(xys), _np(conics), col.numpy(), np.ones((N, 1)), H,
                                           W, offs, ids, want_margin=True)
    ok = margin > 1e-4
    assert ok.mean() > 0.995
    assert np.abs(_np(img) - ref)[ok].max() <= IMG_ATOL
    rg = oracle.rasterize_sum_bwd(_np(xys), _np(conics), col.numpy(), np.ones((N, 1)), H, W, offs,
                                  ids, v_out.numpy())
    for leaf, r, name in zip(leaves, rg[:3], ("xys", "conics", "colors")):
        _assert_grad(_np(leaf.grad), r, name)


def test_4k_frame_properties(ops, oracle):
    """BASELINE config 5 size (3840x2160, 200k): size-independent properties instead of a full
    oracle pass -- linearity in the colours, list bookkeeping, and an oracle check of a band."""
    H, W, N = 2160, 3840, 200000
    means, L, col, op = synth.cholesky_inputs(N, seed=1)
    tb = oracle.tile_bounds(H, W)
    proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    opa = torch.ones(N, 1, device=DEV)
    c1, c2 = col.to(DEV), torch.rand(N, 3, generator=torch.Generator().manual_seed(9)).to(DEV)
    a = ops.rasterize_gaussians_sum(*proj, c1, opa, H, W, 16, 16)
    b = ops.rasterize_gaussians_sum(*proj, c2, opa, H, W, 16, 16)
    ab = ops.rasterize_gaussians_sum(*proj, c1 + c2, opa, H, W, 16, 16)
    assert float((a + b - ab).abs().max()) <= 2e-5
    ws = ops.get_workspace(DEV)
    assert int(ws.bins()[0][-1]) == int(proj[4].sum().item())
    # oracle on the top 64 rows only (same lists restricted to those tile rows)
    Hb = 64
    xys, radii, conics = _np(proj[0]), _np(proj[2]), _np(proj[3])
    offs, ids = oracle.bin_tiles(xys, radii, H, W)
    nb = tb[0] * (Hb // 16)
    ref, margin = oracle.rasterize_sum_fwd(xys, conics, col.numpy(), np.ones((N, 1)), Hb, W,
                                           offs[:nb + 1], ids[:offs[nb]], want_margin=True)
    ok = margin > 1e-4
    assert np.abs(_np(a[:Hb]) - ref)[ok].max() <= IMG_ATOL


@pytest.mark.parametrize("form", [0, 4])
@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_tile_row_bands_equal_the_unsharded_frame(ops, oracle, kind, form):
    """Tile-row sharding on one device: every rank's band, rendered in turn, is bit-identical to
    the same rows of the unsharded frame (same lists, same order), and the per-band partial
    gradients sum to the unsharded gradients.  form 4: the bands fill their lists through cells
    of tiles (the form large frames take by themselves), the unsharded frame does not."""
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FramePipeline
    from gaussianimage_amd.sharding import band_pixels, band_plan
    H, W, N = 200, 176, 4000
    if kind == "cholesky":
        means, a, col, op = synth.cholesky_inputs(N, seed=4, footprint=2.0)
        b = None
    else:
        means, a, b, col, op = synth.rs_inputs(N, seed=4, footprint=2.0)
    args = [None if t is None else t.to(DEV).contiguous() for t in (means, a, b, col)]
    args.append(op.reshape(-1).to(DEV))
    full = FramePipeline(N, H, W, kind=kind, device=DEV)
    img = full.forward(*args).clone()
    v_img = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    ref_grads = [g.clone() for g in full.backward(*args, v_img) if g is not None]
    for world in (2, 3, 8):
        plan = band_plan(full.tiles_y, world)
        sums = None
        for rank in range(world):
            pipe = FramePipeline(N, H, W, kind=kind, device=DEV, rows=plan[rank])
            _lib.check(_lib.lib.gi2d_workspace_set_option(pipe.ws.handle, _lib.OPT_PROJECT_AGG,
                                                          form), "set_option")
            y0, y1 = band_pixels(plan, rank, H)
            band = pipe.forward(*args)
            assert band.shape == (y1 - y0, W, 3)
            assert torch.equal(band, img[y0:y1]), (world, rank)
            grads = [g.clone() for g in pipe.backward(*args, v_img[y0:y1].contiguous())
                     if g is not None]
            sums = grads if sums is None else [s + g for s, g in zip(sums, grads)]
        for s, r in zip(sums, ref_grads):
            assert _maxrel(_np(s), _np(r)) <= 1e-5


def test_asynchronous_overflow_is_loud_and_repaired(ops, oracle):
    """A steady-state (no host synchronisation) frame whose tile lists outgrow the capacity
    sized from earlier frames: the overflowed tiles come out NaN (never silently wrong), a later
    binning call reports it once, and after that the same frame renders correctly."""
    from gaussianimage_amd import _lib
    H, W, N = 64, 64, 4000
    g = torch.Generator().manual_seed(21)
    xys = (torch.rand(N, 2, generator=g) * torch.tensor([W, H])).to(DEV)
    small = torch.ones(N, dtype=torch.int32, device=DEV)          # ~1-2 tiles per Gaussian
    big = torch.full((N,), 40, dtype=torch.int32, device=DEV)      # every tile
    conics = torch.tensor([[0.02, 0.0, 0.02]]).repeat(N, 1).to(DEV)
    col = torch.rand(N, 3, generator=g).to(DEV)
    opa = torch.ones(N, 1, device=DEV)
    z = torch.zeros(N, device=DEV)
    nth = torch.zeros(N, dtype=torch.int32, device=DEV)
    # a fresh workspace: the shared one has grown in earlier tests and never shrinks; strict, so
    # that the report of the earlier frame's overflow comes as the exception it is underneath
    torch.cuda.synchronize()
    ops.reset_workspace(DEV)
    ops.get_workspace(DEV).strict = True
    try:
        _overflow_scenario(ops, oracle, xys, small, big, conics, col, opa, z, nth, H, W)
    finally:
        torch.cuda.synchronize()
        ops.reset_workspace(DEV)


def _overflow_scenario(ops, oracle, xys, small, big, conics, col, opa, z, nth, H, W):
    from gaussianimage_amd import _lib

    def render(radii):
        return ops.rasterize_gaussians_sum(xys, z, radii, conics, nth, col, opa, H, W, 16, 16)

    render(small)                       # learns the shape synchronously, capacity ~2x its lists
    img = render(big.clone())           # same shape, asynchronous: lists 10x longer
    assert torch.isnan(img).any(), "overflowed tiles must be NaN"
    reported = 0
    ok_img = None
    for _ in range(5):                  # each overflowed frame is reported once, by a later call
        try:
            ok_img = render(big.clone())
        except _lib.Gi2dError as e:
            assert e.status == _lib.GI2D_E_PREV_OVERFLOW
            reported += 1
    assert 1 <= reported <= 2           # the frame above, and the one issued before the report
    ok_img = render(big.clone())
    offs, ids = oracle.bin_tiles(_np(xys), _np(big), H, W)
    ref = oracle.rasterize_sum_fwd(_np(xys), _np(conics), _np(col), _np(opa), H, W, offs, ids)
    assert not torch.isnan(ok_img).any()
    assert np.abs(_np(ok_img) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_frame_pipeline_equals_operator_path_and_state_errors(ops, oracle):
    """The one-call frame entry (gi2d_render_*_fwd) gives the operators' bits; a rasterize call
    after a projection that refilled the lists, without the binning call, is refused."""
    import ctypes
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N = 112, 144, 3000
    for kind in ("cholesky", "rs"):
        if kind == "cholesky":
            means, a, col, op = synth.cholesky_inputs(N, seed=9, footprint=1.5)
            b = None
        else:
            means, a, b, col, op = synth.rs_inputs(N, seed=9, footprint=1.5)
        dev = [None if t is None else t.to(DEV).contiguous() for t in (means, a, b, col)]
        opa = op.to(DEV)
        tb = oracle.tile_bounds(H, W)
        if kind == "cholesky":
            proj = ops.project_gaussians_2d(dev[0], dev[1], H, W, tb)
        else:
            proj = ops.project_gaussians_2d_scale_rot(dev[0], dev[1], dev[2], H, W, tb)
        ref = ops.rasterize_gaussians_sum(*proj, dev[3], opa, H, W, 16, 16)
        pipe = FramePipeline(N, H, W, kind=kind, device=DEV)
        for _ in range(3):  # first frame learns the shape synchronously, later ones do not
            img = pipe.forward(dev[0], dev[1], dev[2], dev[3], opa.reshape(-1))
            assert torch.equal(img, ref)
        run = pipe.bind(dev[0], dev[1], dev[2], dev[3], opa.reshape(-1))
        assert torch.equal(run(), ref)
        for out, got in zip(proj, (pipe.xys, pipe.depths, pipe.radii, pipe.conics,
                                   pipe.num_tiles_hit)):
            assert torch.equal(out, got)
        # projection with the workspace refills the lists; rasterizing before gi2d_bin_tiles
        pipe.project(dev[0], dev[1], dev[2])
        rc = _lib.lib.gi2d_rasterize_sum_fwd(
            pipe.ws.handle, ops._ptr(pipe.xys), ops._ptr(pipe.conics), ops._ptr(dev[3]),
            ops._ptr(opa), None, N, H, W, ctypes.byref(pipe.conv), ops._ptr(pipe.image),
            ops._stream(torch.device(DEV)))
        assert rc == _lib.GI2D_E_STATE
        pipe.bin()
        assert torch.equal(pipe.rasterize(dev[3], opa.reshape(-1)), ref)


@pytest.mark.parametrize("H,W,N,seed,fp,sets", [(80, 96, 512, 1, 1.0, 3), (33, 50, 300, 2, 2.0, 2),
                                                (64, 64, 3000, 5, 6.0, 3), (48, 64, 64, 0, 1.0, 1)])
def test_multi_set_pass_equals_separate_passes(ops, oracle, H, W, N, seed, fp, sets):
    """The one-launch rendering of several colour sets (the model's three passes,
    gaussianimage_cholesky.py:81-97) gives, bit for bit, the images of separate calls, both as
    the first pass after a binning (sorting variant) and as a later one; the oracle confirms
    the values; gradients are those of the separate calls."""
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, seed, fp)
    g = torch.Generator().manual_seed(seed + 7)
    op = (0.2 + 0.8 * torch.rand(N, 1, generator=g)).to(DEV)
    cols = [col.to(DEV), (torch.rand(N, 3, generator=g) * 0.5).to(DEV),
            torch.ones(N, 3, device=DEV)][:sets]
    xys, depths, radii, conics, nth = proj
    single = [ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, c, op, H, W, 16, 16)
              for c in cols]
    later = ops.rasterize_gaussians_sum_multi(xys, depths, radii, conics, nth, cols, op, H, W)
    ws.bin(xys, radii, H, W, force=True)  # fresh lists: the multi pass sorts them itself
    first = ops.rasterize_gaussians_sum_multi(xys, depths, radii, conics, nth, cols, op, H, W)
    assert len(later) == len(first) == sets
    for s in range(sets):
        assert torch.equal(later[s], single[s]), s
        assert torch.equal(first[s], single[s]), s
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    for s in range(sets):
        ref, margin = oracle.rasterize_sum_fwd(_np(xys), _np(conics), _np(cols[s]), _np(op), H, W,
                                               offs, ids, want_margin=True)
        ok = margin > 1e-4
        # (the dense case piles hundreds of Gaussians on a pixel: values of ~100, whose float
        # spacing alone is 8e-6, so the bound is taken relative to the image's scale there)
        assert np.abs(_np(first[s]) - ref)[ok].max() <= IMG_ATOL * max(1.0, np.abs(ref).max())
    # gradients: sum over the sets of the single-set gradients
    v = [torch.randn(H, W, 3, generator=g).to(DEV) for _ in range(sets)]

    def grads(multi):
        leaves = [xys.detach().clone().requires_grad_(True),
                  conics.detach().clone().requires_grad_(True), op.clone().requires_grad_(True)]
        cl = [c.clone().requires_grad_(True) for c in cols]
        if multi:
            imgs = ops.rasterize_gaussians_sum_multi(leaves[0], depths, radii, leaves[1], nth, cl,
                                                     leaves[2], H, W)
        else:
            imgs = [ops.rasterize_gaussians_sum(leaves[0], depths, radii, leaves[1], nth, c,
                                                leaves[2], H, W, 16, 16) for c in cl]
        sum((i * w).sum() for i, w in zip(imgs, v)).backward()
        return [t.grad for t in leaves + cl]

    for a, b in zip(grads(True), grads(False)):
        assert a.shape == b.shape
        assert _maxrel(_np(a), _np(b)) <= 1e-6


def test_multi_set_argument_errors(ops, oracle):
    H, W, N = 32, 32, 40
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, 0, 1.0)
    c, o = col.to(DEV), op.to(DEV)
    with pytest.raises(ValueError):
        ops.rasterize_gaussians_sum_multi(*proj, [], o, H, W)
    with pytest.raises(ValueError):
        ops.rasterize_gaussians_sum_multi(*proj, [c] * 4, o, H, W)
    with pytest.raises(ValueError):
        ops.rasterize_gaussians_sum_multi(*proj, [c, c[:, :2]], o, H, W)
    with pytest.raises(RuntimeError):
        ops.rasterize_gaussians_sum_multi(*[p.cpu() for p in proj], [col], op, H, W)


def test_harness_fused_passes_equal_the_three_pass_forward(ops, oracle):
    from gaussianimage_amd._testing.harness import CholeskyImageModel
    torch.manual_seed(3)
    a = CholeskyImageModel(2000, 96, 128, DEV, passes=3)
    b = CholeskyImageModel(2000, 96, 128, DEV, passes="fused")
    b.load_state_dict(a.state_dict())
    b.random_colors = a.random_colors
    with torch.no_grad():
        oa, ob = a(), b()
    assert set(oa) == set(ob) == {"render", "gauss_render", "alpha_map", "final_opacities"}
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k


@pytest.mark.parametrize("H,W,N,seed,fp", [(80, 96, 512, 1, 1.0), (33, 50, 300, 2, 2.0),
                                           (64, 64, 3000, 5, 6.0), (48, 48, 9000, 11, 8.0)])
def test_pixels_per_lane_variants_give_the_same_bits(ops, oracle, H, W, N, seed, fp):
    """The forward rasteriser's workgroup shapes (4, 2, 1 waves per tile) are interchangeable:
    same image bits as the sorting first pass and as a later pass, same finalised lists --
    including lists longer than one batch of the small workgroups (64 / 128 entries) and longer
    than the LDS sort buffer."""
    from gaussianimage_amd.pipeline import FramePipeline
    means, L, col, op = (t.to(DEV) for t in synth.cholesky_inputs(N, seed=seed, footprint=fp))
    op = (0.2 + 0.8 * torch.rand(N, 1, generator=torch.Generator().manual_seed(seed))).to(DEV)
    pipe = FramePipeline(N, H, W, device=DEV, workspace=ops.Workspace(DEV))
    # (the very first frame of a shape sizes the lists from a plain binning of every box tile;
    # from the second on the projection fills them, leaving out tiles the support cannot touch)
    pipe.forward(means, L, None, col, op)
    results = {}
    for ppl in (1, 2, 4):
        pipe.ws.set_raster_ppl(ppl)
        first = pipe.forward(means, L, None, col, op).clone()
        offs, ids = pipe.ws.bins()
        later = pipe.rasterize(col, op).clone()
        results[ppl] = (first, later, offs.clone(), ids.clone())
    pipe.ws.set_raster_ppl(0)
    base = results[1]
    assert torch.equal(base[0], base[1])
    lens = np.diff(base[2].numpy())
    if N >= 3000:
        assert lens.max() > 256  # the long-list paths are exercised
    for ppl in (2, 4):
        for a, b in zip(results[ppl], base):
            assert torch.equal(a, b), ppl
    with pytest.raises(Exception):
        pipe.ws.set_raster_ppl(3)


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_fit_forward_is_the_same_for_every_pixels_per_lane(ops, oracle, kind):
    """Loss epilogue variant: image gradient, per-tile loss partials and rendered image do not
    depend on the workgroup shape."""
    from gaussianimage_amd.fit import FusedFitter
    H, W, N = 96, 112, 1500
    gt = synth.target_image(H, W).to(DEV)
    got = {}
    for ppl in (1, 2, 4):
        torch.manual_seed(5)
        fit = FusedFitter(N, H, W, kind=kind, device=DEV, keep_image=True)
        fit.ws.set_raster_ppl(ppl)
        fit.forward(gt)
        got[ppl] = (fit.v_image.clone(), fit.loss_partials.clone(), fit.image.clone())
    for ppl in (2, 4):
        for a, b in zip(got[ppl], got[1]):
            assert torch.equal(a, b), ppl


def test_masked_opacities_and_their_gradient(ops, oracle):
    """The pruning variant of the model multiplies the opacities by a (partly zero) mask that
    needs a gradient (gaussianimage_cholesky_wMask.py:221-226): zero-opacity Gaussians
    contribute nothing, and v_opacity is right for every Gaussian, masked ones included."""
    H, W, N = 80, 96, 700
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, 9, 2.0)
    g = torch.Generator().manual_seed(9)
    mask = (torch.rand(N, 1, generator=g) > 0.4).float()
    soft = torch.rand(N, 1, generator=g)
    opac = torch.where(torch.rand(N, 1, generator=g) > 0.5, mask, mask * soft)
    xys, depths, radii, conics, nth = proj
    o = opac.to(DEV).requires_grad_(True)
    c = col.to(DEV).requires_grad_(True)
    img = ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, c, o, H, W, 16, 16)
    v_out = torch.randn(H, W, 3, generator=g)
    (img * v_out.to(DEV)).sum().backward()
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    ref, margin = oracle.rasterize_sum_fwd(_np(xys), _np(conics), col.numpy(), opac.numpy(), H, W,
                                           offs, ids, want_margin=True)
    ok = margin > 1e-4
    assert np.abs(_np(img) - ref)[ok].max() <= IMG_ATOL
    r = oracle.rasterize_sum_bwd(_np(xys), _np(conics), col.numpy(), opac.numpy(), H, W, offs, ids,
                                 v_out.numpy())
    _assert_grad(_np(c.grad), r[2], "colors")
    _assert_grad(_np(o.grad), r[3].reshape(N, 1), "opacity")
    dead = (opac == 0).reshape(-1).numpy()
    assert dead.sum() > 100 and not _np(c.grad)[dead].any()


@pytest.mark.parametrize("H,W,N", [(768, 1020, 50000), (2160, 3840, 200000)])
def test_full_size_backward_properties(ops, oracle, H, W, N):
    """BASELINE configs 4 and 5 sizes, forward + backward, through properties that need no
    oracle pass: the image is linear in the colours, so (a) the colour gradient does not depend
    on the colours (bit for bit), (b) <v_colors, d> equals the change of the loss along any
    colour direction d, (c) the Gaussians split into two halves render to images that add up to
    the full frame, with gradients that are those of the full frame."""
    means, L, col, op = synth.cholesky_inputs(N, seed=2)
    tb = oracle.tile_bounds(H, W)
    g = torch.Generator().manual_seed(3)
    proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    xys, depths, radii, conics, nth = proj
    opa = (0.3 + 0.7 * torch.rand(N, 1, generator=g)).to(DEV)
    v_out = torch.randn(H, W, 3, generator=g).to(DEV)

    def run(colors, keep=None):
        x = xys.detach().clone().requires_grad_(True)
        c = colors.clone().requires_grad_(True)
        o = opa.clone() if keep is None else opa * keep
        img = ops.rasterize_gaussians_sum(x, depths, radii, conics, nth, c, o, H, W, 16, 16)
        loss = (img * v_out).sum()
        loss.backward()
        return img.detach(), float(loss.detach()), x.grad, c.grad

    c1 = col.to(DEV)
    c2 = torch.rand(N, 3, generator=g).to(DEV)
    img1, l1, gx1, gc1 = run(c1)
    img2, l2, gx2, gc2 = run(c2)
    assert torch.equal(gc1, gc2)                                        # (a)
    d = c2 - c1
    lhs, rhs = l2 - l1, float((gc1.double() * d.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0)        # (b)
    even = (torch.arange(N, device=DEV) % 2 == 0).float().view(N, 1)
    img_e, _, gx_e, gc_e = run(c1, even)
    img_o, _, gx_o, gc_o = run(c1, 1 - even)
    assert float((img_e + img_o - img1).abs().max()) <= 2e-5            # (c)
    m = even.bool().view(-1)
    # (equal up to the order of the partial sums, which follows the tile's total work)
    assert _maxrel(_np(gc_e[m]), _np(gc1[m])) <= 2e-6
    assert _maxrel(_np(gc_o[~m]), _np(gc1[~m])) <= 2e-6
    assert not gc_e[~m].any() and not gx_e[~m].any()
    assert _maxrel(_np(gx_e + gx_o), _np(gx1)) <= 1e-5


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_fused_projection_lists_only_tiles_the_support_can_touch(ops, oracle, kind):
    """With the opacities at hand the one-call frame leaves out (tile, Gaussian) pairs whose
    support rectangle on that tile is empty: its lists are a subset of the reference lists
    (every tile of the radius box), every omitted pair is one the oracle gives no contribution
    on any pixel, image and gradients are unchanged, and with culling off the lists are the
    reference lists again."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N = 112, 144, 3000
    g = torch.Generator().manual_seed(4)
    if kind == "cholesky":
        means, a, col, _ = synth.cholesky_inputs(N, seed=9, footprint=1.5)
        b = None
    else:
        means, a, b, col, _ = synth.rs_inputs(N, seed=9, footprint=1.5)
    opa = torch.where(torch.rand(N, generator=g) > 0.3, torch.rand(N, generator=g),
                      torch.zeros(N)).to(DEV)  # some low, some zero opacities
    dev = [None if t is None else t.to(DEV).contiguous() for t in (means, a, b, col)]
    pipe = FramePipeline(N, H, W, kind=kind, device=DEV, workspace=ops.Workspace(DEV))
    pipe.forward(*dev, opa)                      # sizes the lists (plain binning)
    img = pipe.forward(*dev, opa).clone()        # lists filled by the projection, culled
    offs, ids = pipe.ws.bins()
    r_offs, r_ids = oracle.bin_tiles(_np(pipe.xys), _np(pipe.radii), H, W)
    assert int(offs[-1]) < int(r_offs[-1]) * 0.9
    xys, conics = _np(pipe.xys), _np(pipe.conics)
    op_np = _np(opa)
    tiles_x = (W + 15) // 16
    dropped = 0
    for t in range(len(r_offs) - 1):
        mine = set(ids[offs[t]:offs[t + 1]].tolist())
        full = r_ids[r_offs[t]:r_offs[t + 1]]
        assert mine <= set(full.tolist())
        gone = np.array(sorted(set(full.tolist()) - mine), dtype=np.int64)
        if len(gone) == 0:
            continue
        dropped += len(gone)
        # alpha of every dropped Gaussian on every pixel of the tile, in float64
        ty, tx = divmod(t, tiles_x)
        ys, xs = np.mgrid[ty * 16:min(ty * 16 + 16, H), tx * 16:min(tx * 16 + 16, W)]
        dx = xys[gone, 0][:, None, None] - xs[None]
        dy = xys[gone, 1][:, None, None] - ys[None]
        ca, cb, cc = (conics[gone, k].astype(np.float64)[:, None, None] for k in range(3))
        sigma = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
        alpha = op_np[gone].astype(np.float64)[:, None, None] * np.exp(-sigma)
        assert (alpha < 1.0 / 255.0).all()
    assert dropped == int(r_offs[-1]) - int(offs[-1])
    v = torch.randn(H, W, 3, generator=g).to(DEV)
    grads = [t.clone() for t in pipe.backward(*dev, opa, v) if t is not None]
    pipe.ws.set_cull(False)
    img_full = pipe.forward(*dev, opa).clone()
    offs2, ids2 = pipe.ws.bins()
    assert np.array_equal(offs2.numpy(), r_offs) and np.array_equal(ids2.numpy(), r_ids)
    assert torch.equal(img_full, img)
    grads_full = [t for t in pipe.backward(*dev, opa, v) if t is not None]
    for a_, b_ in zip(grads, grads_full):
        assert _maxrel(_np(a_), _np(b_)) <= 2e-6
    pipe.ws.set_cull(True)


@pytest.fixture
def other_conventions(ops):
    """The other plausible reading of the operators the reference does not contain: pixel
    centres at +0.5 with the matching -0.5 bias, the 0.999 alpha cap, background added."""
    conv = dict(pixel_offset=0.5, center_bias=-0.5, alpha_cap=0.999, bg_weight=1.0)
    before = ops.get_conventions()
    ops.set_conventions(**conv)
    yield conv
    ops.set_conventions(**before)


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_runtime_conventions_reach_every_kernel(ops, oracle, other_conventions, kind):
    """Projection, per-tile and per-list culling, forward and backward all follow the runtime
    convention fields: operator path and one-call frame against the oracle run with the same
    fields (alpha cap active: opacities up to 1.5)."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N = 90, 120, 1500
    conv = oracle.default_conv(**other_conventions)
    g = torch.Generator().manual_seed(12)
    if kind == "cholesky":
        means, a, col, _ = synth.cholesky_inputs(N, seed=6, footprint=2.0)
        b = None
    else:
        means, a, b, col, _ = synth.rs_inputs(N, seed=6, footprint=2.0)
    opa = (1.5 * torch.rand(N, 1, generator=g))
    bg = torch.tensor([0.2, 0.5, 0.7])
    tb = oracle.tile_bounds(H, W)
    dev = [None if t is None else t.to(DEV) for t in (means, a, b)]
    if kind == "cholesky":
        proj = ops.project_gaussians_2d(dev[0], dev[1], H, W, tb)
        r_proj = oracle.project_cholesky_fwd(means.numpy(), a.numpy(), H, W, conv)
        assert np.array_equal(_np(proj[0]), r_proj[0]) and np.array_equal(_np(proj[2]), r_proj[2])
    else:
        proj = ops.project_gaussians_2d_scale_rot(dev[0], dev[1], dev[2], H, W, tb)
    xys, depths, radii, conics, nth = proj
    leaves = [xys.detach().clone().requires_grad_(True),
              conics.detach().clone().requires_grad_(True), col.to(DEV).requires_grad_(True),
              opa.to(DEV).requires_grad_(True)]
    img = ops.rasterize_gaussians_sum(leaves[0], depths, radii, leaves[1], nth, leaves[2],
                                      leaves[3], H, W, 16, 16, background=bg.to(DEV))
    v_out = torch.randn(H, W, 3, generator=g)
    (img * v_out.to(DEV)).sum().backward()
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    ref, margin = oracle.rasterize_sum_fwd(_np(xys), _np(conics), col.numpy(), opa.numpy(), H, W,
                                           offs, ids, background=bg.numpy(), conv=conv,
                                           want_margin=True)
    ok = margin > 1e-4
    assert ok.mean() > 0.99
    assert np.abs(_np(img) - ref)[ok].max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(ref - bg.numpy()).max()) > 0.1          # the Gaussians are there ...
    assert np.abs(_np(img)[0, 0] - ref[0, 0]).max() <= IMG_ATOL  # ... on top of the background
    rg = oracle.rasterize_sum_bwd(_np(xys), _np(conics), col.numpy(), opa.numpy(), H, W, offs,
                                  ids, v_out.numpy(), conv=conv)
    for name, leaf, r in zip(("xys", "conics", "colors", "opacity"), leaves, rg):
        _assert_grad(_np(leaf.grad), r.reshape(leaf.shape), name)
    # the one-call frame (lists culled with the same pixel offset) gives the operators' bits
    pipe = FramePipeline(N, H, W, kind=kind, device=DEV, workspace=ops.Workspace(DEV))
    pipe.background.copy_(bg)
    args = (*dev, col.to(DEV), opa.to(DEV))
    pipe.forward(*args)
    assert torch.equal(pipe.forward(*args), img.detach())


@pytest.mark.parametrize("H,W,N,fp", [(2160, 3840, 1000000, 1.0), (17, 5000, 20000, 2.0),
                                      (4001, 23, 20000, 2.0), (1, 1, 50, 1.0)])
def test_large_and_odd_shapes(ops, oracle, H, W, N, fp):
    """A million Gaussians on a 4K frame, one-tile-high and one-tile-wide strips, a single
    pixel: the one-call frame equals the operator path bit for bit, the backward's colour
    gradient pairs with the forward (linearity), nothing is NaN."""
    from gaussianimage_amd.pipeline import FramePipeline
    means, L, col, op = (t.to(DEV) for t in synth.cholesky_inputs(N, seed=8, footprint=fp))
    op = op.reshape(-1)
    tb = oracle.tile_bounds(H, W)
    proj = ops.project_gaussians_2d(means, L, H, W, tb)
    ref = ops.rasterize_gaussians_sum(*proj, col, op.reshape(N, 1), H, W, 16, 16)
    pipe = FramePipeline(N, H, W, device=DEV, workspace=ops.Workspace(DEV))
    for _ in range(2):
        img = pipe.forward(means, L, None, col, op)
        assert torch.equal(img, ref)
    assert torch.isfinite(img).all()
    g = torch.Generator().manual_seed(1)
    v = torch.randn(H, W, 3, generator=g).to(DEV)
    grads = pipe.backward(means, L, None, col, op, v)
    assert all(torch.isfinite(t).all() for t in grads if t is not None)
    v_col = grads[3]
    d = torch.rand(N, 3, generator=g).to(DEV)
    img_d = pipe.forward(means, L, None, d, op)
    lhs = float((img_d.double() * v.double()).sum())
    rhs = float((v_col.double() * d.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1.0)


def test_frame_ring_renders_the_same_frames(ops, oracle):
    """Frames issued through the two-deep ring equal the single-stream frames bit for bit,
    whichever slot renders them, with inputs that change from frame to frame."""
    from gaussianimage_amd.pipeline import FramePipeline, FrameRing
    H, W, N = 96, 128, 2500
    pipe = FramePipeline(N, H, W, device=DEV, workspace=ops.Workspace(DEV))
    ring = FrameRing(N, H, W, device=DEV, depth=2)
    frames = []
    for seed in range(5):
        means, L, col, op = (t.to(DEV) for t in synth.cholesky_inputs(N, seed=seed, footprint=1.5))
        img, ev = ring.render(means, L, None, col, op.reshape(-1))
        frames.append((img, ev, (means, L, col, op)))
        if len(frames) >= 2:   # consume the frame before last while the last is in flight
            old_img, old_ev, (m, l, c, o) = frames[-2]
            old_ev.synchronize()
            assert torch.equal(old_img, pipe.forward(m, l, None, c, o.reshape(-1)))
    ring.synchronize()
    img, ev, (m, l, c, o) = frames[-1]
    assert torch.equal(img, pipe.forward(m, l, None, c, o.reshape(-1)))


def test_no_grad_calls_give_the_same_tensors(ops):
    """Under torch.no_grad() (and for inputs that need no gradient) the operators skip the
    autograd bookkeeping; what they return is unchanged."""
    H, W, N = 64, 80, 900
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    means, L, col, op = (t.to(DEV) for t in synth.cholesky_inputs(N, seed=3, footprint=1.5))
    mg, Lg, cg = (t.clone().requires_grad_(True) for t in (means, L, col))
    with_grad = ops.project_gaussians_2d(mg, Lg, H, W, tb)
    img_grad = ops.rasterize_gaussians_sum(*with_grad, cg, op, H, W, 16, 16)
    assert img_grad.requires_grad and with_grad[0].requires_grad
    for ctxmgr in (torch.no_grad(), torch.enable_grad()):
        with ctxmgr:
            a = ops.project_gaussians_2d(mg if ctxmgr.__class__ is torch.no_grad else means,
                                         Lg if ctxmgr.__class__ is torch.no_grad else L, H, W, tb)
            img = ops.rasterize_gaussians_sum(*a, col, op, H, W, 16, 16)
            three = ops.rasterize_gaussians_sum_multi(*a, [col, col * 0.5, torch.ones_like(col)],
                                                      op, H, W)
        assert not img.requires_grad and not a[0].requires_grad
        for x, y in zip(a, with_grad):
            assert torch.equal(x, y.detach())
        assert torch.equal(img, img_grad.detach()) and torch.equal(three[0], img)
    img_grad.sum().backward()   # the recorded path is intact
    assert torch.isfinite(mg.grad).all() and torch.isfinite(cg.grad).all()


@pytest.mark.parametrize("N,footprint", [(400, 1.0), (3000, 1.0), (9000, 1.5)])
def test_one_wave_backward_matches_the_workgroup_backward(ops, oracle, N, footprint):
    """The backward that gives a tile one wave (chosen by itself for sparse frames, forced here
    through the workgroup-shape option) against the 256-lane backward and the oracle: lists of
    a few, of some tens and of more than a wave's worth of ids."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W = 112, 144
    means, L, col, op = (t.to(DEV) for t in synth.cholesky_inputs(N, seed=5, footprint=footprint))
    op = (0.3 + 0.7 * torch.rand(N, 1, device=DEV))
    v_img = torch.randn(H, W, 3, device=DEV)
    grads = {}
    for ppl in (1, 4):
        pipe = FramePipeline(N, H, W, device=DEV, workspace=ops.Workspace(DEV))
        pipe.ws.set_raster_ppl(ppl)
        pipe.forward(means, L, None, col, op.reshape(-1))
        grads[ppl] = [g.clone() for g in pipe.backward(means, L, None, col, op.reshape(-1), v_img)
                      if g is not None]
    for a, b in zip(grads[4], grads[1]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale
    xys, _, radii, conics, _ = oracle.project_cholesky_fwd(means.cpu().numpy(), L.cpu().numpy(),
                                                           H, W)
    offs, ids = oracle.bin_tiles(xys, radii, H, W)
    ref = oracle.rasterize_sum_bwd(xys, conics, col.cpu().numpy(), op.cpu().numpy(), H, W, offs,
                                   ids, v_img.cpu().numpy())
    v_xy_ref, v_conic_ref = ref[0], ref[1]
    v_m, v_L = oracle.project_cholesky_bwd(means.cpu().numpy(), L.cpu().numpy(), H, W, v_xy_ref,
                                           v_conic_ref)
    for got, want in ((grads[4][0], v_m), (grads[4][1], v_L)):
        want = torch.from_numpy(want).to(DEV)
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("N,footprint", [(3000, 1.0), (20000, 2.0)])
def test_spatially_ordered_gaussians_fill_the_same_lists(ops, oracle, N, footprint):
    """Gaussians stored in a spatial order (neighbours in memory are neighbours in the image):
    the lanes of a wave that enter the same tile then take their places from ONE atomic.  The
    lists must come out exactly as the oracle's, through every entry point that fills them."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W = 176, 240
    means, L, col, op = synth.cholesky_inputs(N, seed=11, footprint=footprint)
    cx, cy = 0.5 * W * means[:, 0] + 0.5 * W, 0.5 * H * means[:, 1] + 0.5 * H
    key = (cy / 16).floor() * 1000 + (cx / 16).floor()          # tile of the centre, row-major
    order = torch.argsort(key, stable=True)
    means, L, col, op = (t[order].contiguous().to(DEV) for t in (means, L, col, op))
    xys_o, _, radii_o, conics_o, _ = oracle.project_cholesky_fwd(means.cpu().numpy(),
                                                                 L.cpu().numpy(), H, W)
    offs_o, ids_o = oracle.bin_tiles(xys_o, radii_o, H, W)
    img_o = oracle.rasterize_sum_fwd(xys_o, conics_o, col.cpu().numpy(), op.cpu().numpy(), H, W,
                                     offs_o, ids_o)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    # the operator path: projection fills the lists (full radius boxes, like the oracle)
    proj = ops.project_gaussians_2d(means, L, H, W, tb)
    img = ops.rasterize_gaussians_sum(*proj, col, op, H, W, 16, 16)
    offs, ids = ops.get_workspace(torch.device(DEV)).bins()
    assert np.array_equal(offs.numpy(), offs_o) and np.array_equal(ids.numpy(), ids_o)
    assert float((img.cpu() - torch.from_numpy(img_o)).abs().max()) <= 1e-5 * max(1.0, float(img.max()))
    # binning from caller-made xys / radii (gi2d_bin_tiles)
    ws = ops.Workspace(DEV)
    ws.bin(proj[0].clone(), proj[2].clone(), H, W, force=True)
    offs2, ids2 = ws.bins()
    assert np.array_equal(offs2.numpy(), offs_o) and np.array_equal(ids2.numpy(), ids_o)
    # the one-call frame (culled lists): same image bits as the operator path
    pipe = FramePipeline(N, H, W, device=DEV, workspace=ops.Workspace(DEV))
    assert torch.equal(pipe.forward(means, L, None, col, op.reshape(-1)), img)
    assert torch.equal(pipe.forward(means, L, None, col, op.reshape(-1)), img)


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_shapes_against_the_oracle(ops, oracle, seed):
    """A sweep over random image sizes (down to one pixel, mostly not multiples of 16), counts,
    footprints and opacities (some zero, some below the alpha cut): lists bit for bit, image and
    gradients within the stated tolerances."""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    N = int(rng.integers(1, 4000))
    fp = float(rng.uniform(0.5, 4.0))
    means, L, col, _ = synth.cholesky_inputs(N, seed=seed, footprint=fp)
    op = torch.from_numpy(rng.uniform(0.0, 1.0, (N, 1)).astype(np.float32))
    op[rng.integers(0, N, max(1, N // 10))] = 0.0
    op[rng.integers(0, N, max(1, N // 10))] = 0.003      # below 1/255: can never contribute
    tb = oracle.tile_bounds(H, W)
    leaves = [means.to(DEV).requires_grad_(True), L.to(DEV).requires_grad_(True),
              col.to(DEV).requires_grad_(True), op.to(DEV).requires_grad_(True)]
    proj = ops.project_gaussians_2d(leaves[0], leaves[1], H, W, tb)
    img = ops.rasterize_gaussians_sum(*proj, leaves[2], leaves[3], H, W, 16, 16)
    offs, ids = ops.get_workspace(torch.device(DEV)).bins()
    ref = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    for k in (0, 2, 3, 4):
        assert np.array_equal(_np(proj[k]), ref[k]), k
    r_offs, r_ids = oracle.bin_tiles(ref[0], ref[2], H, W)
    assert np.array_equal(offs.numpy(), r_offs) and np.array_equal(ids.numpy(), r_ids)
    r_img, margin = oracle.rasterize_sum_fwd(ref[0], ref[3], col.numpy(), op.numpy(), H, W,
                                             r_offs, r_ids, want_margin=True)
    ok = margin > 1e-4
    err = np.abs(_np(img) - r_img)
    if ok.any():
        assert err[ok].max() <= IMG_ATOL * max(1.0, float(np.abs(r_img).max()))
    assert err.max() <= 3.0 * (1 / 255) * max(1.0, float(col.max()))
    # (a pair sitting on the alpha cut counts in one implementation and not in the other: such
    # pixels -- the oracle names them -- get no image gradient, so the pair counts in neither)
    v_out = torch.from_numpy(rng.standard_normal((H, W, 3)).astype(np.float32) * ok[..., None])
    (img * v_out.to(DEV)).sum().backward()
    r_bwd = oracle.rasterize_sum_bwd(ref[0], ref[3], col.numpy(), op.numpy(), H, W, r_offs, r_ids,
                                     v_out.numpy())
    r_m, r_L = oracle.project_cholesky_bwd(means.numpy(), L.numpy(), H, W, r_bwd[0], r_bwd[1])
    for name, leaf, r in (("means", leaves[0], r_m), ("L", leaves[1], r_L),
                          ("colors", leaves[2], r_bwd[2]), ("opacity", leaves[3], r_bwd[3])):
        r = np.asarray(r).reshape(leaf.shape)
        if np.abs(r).max() > 0:
            _assert_grad(_np(leaf.grad), r, (name, H, W, N, fp))
        else:
            assert not _np(leaf.grad).any(), name


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_bound_frame_call_equals_forward(ops, oracle, kind):
    """FramePipeline.bind() (gi2d_render_frame, arguments in one struct) renders what forward()
    (gi2d_render_*_fwd, twenty arguments) renders, with and without the instrumentation events."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N = 90, 130, 2000
    if kind == "cholesky":
        means, cov_a, col, op = (t.to(DEV) for t in synth.cholesky_inputs(N, seed=2))
        cov_b = None
    else:
        means, cov_a, cov_b, col, op = (t.to(DEV) for t in synth.rs_inputs(N, seed=2))
    op = op.reshape(-1)
    a = FramePipeline(N, H, W, kind=kind, device=DEV, workspace=ops.Workspace(DEV))
    b = FramePipeline(N, H, W, kind=kind, device=DEV, workspace=ops.Workspace(DEV))
    want = a.forward(means, cov_a, cov_b, col, op).clone()
    run = b.bind(means, cov_a, cov_b, col, op)
    assert torch.equal(run(), want)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in ev:
        e.record()
    assert torch.equal(run(ev), want)
    torch.cuda.synchronize()
    assert ev[0].elapsed_time(ev[1]) > 0
    assert torch.equal(run(), want)
    for x, y in ((a.xys, b.xys), (a.radii, b.radii), (a.conics, b.conics)):
        assert torch.equal(x, y)


def test_overflow_with_shared_atomics_is_reported_and_repaired(ops, oracle):
    """Spatially ordered Gaussians (lanes share list atomics) against a capacity that is too
    small: the overflow is reported, never written past a segment, and the grown workspace bins
    exactly."""
    from gaussianimage_amd import _lib
    H, W, N = 64, 96, 6000
    means, L, col, op = synth.cholesky_inputs(N, seed=8, footprint=2.0)
    key = ((0.5 * H * means[:, 1] + 0.5 * H) / 16).floor() * 100 + ((0.5 * W * means[:, 0] + 0.5 * W) / 16).floor()
    order = torch.argsort(key, stable=True)
    means, L = means[order].contiguous(), L[order].contiguous()
    tb = oracle.tile_bounds(H, W)
    ws = ops.Workspace(DEV)
    ws.reserve(64, 24)
    proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
    with pytest.raises(_lib.Gi2dError) as ei:
        ws.bin(proj[0], proj[2], H, W, flags=_lib.BIN_SYNC | _lib.BIN_NOGROW, force=True)
    assert ei.value.status == _lib.GI2D_E_CAPACITY
    ws.bin(proj[0], proj[2], H, W, flags=_lib.BIN_SYNC, force=True)
    offs, ids = ws.bins()
    r_offs, r_ids = oracle.bin_tiles(_np(proj[0]), _np(proj[2]), H, W)
    assert np.diff(r_offs).max() > 64
    assert np.array_equal(ids.numpy(), r_ids) and np.array_equal(offs.numpy(), r_offs)


def test_two_backwards_on_one_binning_with_different_opacities(ops, oracle):
    """One projection, one binning, two backward calls whose opacities differ: a (Gaussian, tile)
    pair the first call reaches and the second does not (its support shrank with the opacity)
    must contribute nothing to the second call's gradients -- every backward launch carries its
    own stamp, so what an earlier launch left in a Gaussian's slots never counts."""
    H, W, N = 96, 128, 1500
    means, L, col, op, proj, ws = _project_and_bin(ops, oracle, H, W, N, 12, 2.5)
    xys, depths, radii, conics, nth = proj
    v_out = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(12))
    offs, ids = oracle.bin_tiles(_np(xys), _np(radii), H, W)
    gen0 = ws.info()["generation"]
    for opacity in (1.0, 0.01, 0.3, 1.0):
        opa = torch.full((N, 1), opacity)
        leaves = [xys.detach().requires_grad_(True), conics.detach().requires_grad_(True),
                  col.to(DEV).requires_grad_(True), opa.to(DEV).requires_grad_(True)]
        # (the leaves share xys' storage and version: the workspace keeps its one binning)
        img = ops.rasterize_gaussians_sum(leaves[0], depths, radii, leaves[1], nth, leaves[2],
                                          leaves[3], H, W, 16, 16)
        (img * v_out.to(DEV)).sum().backward()
        ref = oracle.rasterize_sum_bwd(_np(xys), _np(conics), col.numpy(), opa.numpy(), H, W, offs,
                                       ids, v_out.numpy())
        for name, leaf, r in zip(("xys", "conics", "colors", "opacity"), leaves, ref):
            _assert_grad(_np(leaf.grad), r.reshape(leaf.shape), (name, opacity))
    assert ws.info()["generation"] == gen0, "all four passes ran on the one binning"


def test_backward_stamps_wrap_without_stale_sums(ops, oracle):
    """The stamp counter is positioned just before its wrap (a testing option of the C ABI): the
    launches across the wrap give the gradients of the launches before it."""
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N = 80, 96, 900
    means, L, col, op = synth.cholesky_inputs(N, seed=13, footprint=2.0)
    args = [means.to(DEV), L.to(DEV), None, col.to(DEV), op.reshape(-1).to(DEV)]
    low = list(args)
    low[4] = torch.full((N,), 0.02, device=DEV)
    pipe = FramePipeline(N, H, W, device=DEV)
    v_img = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(13)).to(DEV)
    pipe.forward(*args)
    want = [g.clone() for g in pipe.backward(*args, v_img) if g is not None]
    want_low = [g.clone() for g in pipe.backward(*low, v_img) if g is not None]
    _lib.check(_lib.lib.gi2d_workspace_set_option(pipe.ws.handle, _lib.OPT_BWD_SERIAL,
                                                  2 ** 31 - 3), "set_option")
    for k in range(4):  # stamps INT_MAX-1, INT_MAX, 1 (slots cleared), 2
        a = low if k % 2 else args
        got = [g for g in pipe.backward(*a, v_img) if g is not None]
        for g, w in zip(got, want_low if k % 2 else want):
            assert torch.equal(g, w), k


def test_two_streams_through_the_operators_do_not_share_lists(ops, oracle):
    """Two frames on two streams, their projection / rasterisation / backward calls interleaved
    from one host thread: every stream has its own operator workspace, so each frame gets the
    image and gradients it gets alone on the default stream."""
    H, W = 96, 128
    tb = oracle.tile_bounds(H, W)
    frames = []
    for seed, N, fp in ((21, 2500, 2.0), (22, 1200, 1.0)):
        means, L, col, op = synth.cholesky_inputs(N, seed=seed, footprint=fp)
        v = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(seed))
        frames.append([t.to(DEV) for t in (means, L, col, op, v)])

    def leaves(f):
        return [t.clone().requires_grad_(True) for t in f[:3]]

    # alone, one after the other, on the default stream
    want = []
    for f in frames:
        m, l, c = leaves(f)
        img = ops.rasterize_gaussians_sum(*ops.project_gaussians_2d(m, l, H, W, tb), c, f[3], H, W,
                                          16, 16)
        (img * f[4]).sum().backward()
        want.append([img.detach().clone(), m.grad.clone(), l.grad.clone(), c.grad.clone()])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    assert ops.get_workspace(DEV) is ops.get_workspace(DEV)
    with torch.cuda.stream(streams[0]):
        ws0 = ops.get_workspace(DEV)
    with torch.cuda.stream(streams[1]):
        assert ops.get_workspace(DEV) is not ws0
    lv = [leaves(f) for f in frames]
    proj, imgs = [None, None], [None, None]
    for rep in range(3):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                proj[k] = ops.project_gaussians_2d(lv[k][0], lv[k][1], H, W, tb)
        for k in (1, 0):
            with torch.cuda.stream(streams[k]):
                imgs[k] = ops.rasterize_gaussians_sum(*proj[k], lv[k][2], frames[k][3], H, W, 16, 16)
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                for t in lv[k]:
                    t.grad = None
                (imgs[k] * frames[k][4]).sum().backward()
    torch.cuda.synchronize()
    for k in (0, 1):
        got = [imgs[k].detach(), lv[k][0].grad, lv[k][1].grad, lv[k][2].grad]
        for g, w in zip(got, want[k]):
            assert torch.equal(g, w), k


def test_late_overflow_report_does_not_stop_the_operators(ops, oracle):
    """The operators swallow the report of an EARLIER frame's overflow: they count it and warn
    on every occurrence (with the occurrence number and the lists' generation), and the image of
    the call that carries the report is complete; strict=True brings the exception back."""
    import warnings
    from gaussianimage_amd import _lib
    H, W, N = 64, 64, 1500
    tb = oracle.tile_bounds(H, W)
    z = torch.zeros(N, device=DEV)
    nth = torch.zeros(N, dtype=torch.int32, device=DEV)
    col, opa = torch.rand(N, 3, device=DEV), torch.ones(N, 1, device=DEV)
    g = torch.Generator().manual_seed(3)
    xys = (torch.rand(N, 2, generator=g) * torch.tensor([W, H])).to(DEV)
    conics = torch.tensor([[0.5, 0.0, 0.5]]).repeat(N, 1).to(DEV)
    small = torch.full((N,), 2, dtype=torch.int32, device=DEV)
    big = torch.full((N,), 60, dtype=torch.int32, device=DEV)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):     # a stream of its own, and a fresh workspace on it
        ops.reset_workspace(DEV)
        ws = ops.get_workspace(DEV)
        for _ in range(3):
            ops.rasterize_gaussians_sum(xys.clone(), z, small, conics, nth, col, opa, H, W, 16, 16)
        torch.cuda.synchronize()
        img = ops.rasterize_gaussians_sum(xys.clone(), z, big, conics, nth, col, opa, H, W, 16, 16)
        torch.cuda.synchronize()
        assert torch.isnan(img).any()          # that frame's overflowed tiles are marked
        before = ws.late_overflows
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            # (the report travels device -> pinned host memory with the NEXT launch: it reaches
            # the host one or two calls after the frame it is about)
            for _ in range(3):
                img = ops.rasterize_gaussians_sum(xys.clone(), z, big, conics, nth, col, opa, H,
                                                  W, 16, 16)
                torch.cuda.synchronize()
        assert 1 <= ws.late_overflows - before <= 2 and torch.isfinite(img).all()
        told = [str(w.message) for w in caught if "overflowed a tile list" in str(w.message)]
        assert len(told) == ws.late_overflows - before     # one warning per occurrence
        assert all("generation" in t and "NaN" in t for t in told)


@pytest.mark.parametrize("compiled", [True, False])
def test_recycled_projection_outputs_are_not_taken_for_the_projected_ones(ops, oracle, compiled):
    """The projection prefills the tile lists for ITS outputs.  If those outputs die unrasterised
    and the allocator hands their addresses to unrelated xys / radii tensors of the same shape,
    the rasteriser must bin the new tensors, not reuse the prefilled lists."""
    if compiled and not ops.compiled_binding():
        pytest.skip("compiled binding not loaded")
    prev = ops.use_compiled(compiled)
    try:
        H, W, N = 64, 80, 700
        tb = oracle.tile_bounds(H, W)
        means, L, col, op = synth.cholesky_inputs(N, seed=21)
        other_means, other_L, _, _ = synth.cholesky_inputs(N, seed=22, footprint=2.0)
        want = oracle.project_cholesky_fwd(other_means.numpy(), other_L.numpy(), H, W)
        col_d, op_d = col.to(DEV), op.to(DEV)
        for _ in range(3):
            proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
            torch.cuda.synchronize()
            del proj                      # never rasterised; its blocks go back to the allocator
            xys = torch.from_numpy(want[0]).to(DEV)          # fresh tensors, version 0
            depths = torch.zeros(N, device=DEV)
            radii = torch.from_numpy(want[2]).to(DEV)
            conics = torch.from_numpy(want[3]).to(DEV)
            nth = torch.from_numpy(want[4]).to(DEV)
            img = ops.rasterize_gaussians_sum(xys, depths, radii, conics, nth, col_d, op_d, H, W,
                                              16, 16)
            offs, ids = ops.get_workspace(DEV).bins()
            r_offs, r_ids = oracle.bin_tiles(want[0], want[2], H, W)
            assert np.array_equal(offs.numpy(), r_offs) and np.array_equal(ids.numpy(), r_ids)
            ref, margin = oracle.rasterize_sum_fwd(want[0], want[3], col.numpy(), op.numpy(), H, W,
                                                   r_offs, r_ids, want_margin=True)
            assert np.abs(_np(img) - ref)[margin > 1e-4].max() <= IMG_ATOL
            del xys, depths, radii, conics, nth, img
    finally:
        ops.use_compiled(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("compiled", [True, False])
def test_other_channel_counts_and_the_accumulated_weight(compiled):
    """colors[N, D] for D != 3 and return_alpha=True run through the multi-set rasteriser: every
    channel is the bits of the same channel rendered in a 3-channel call, alpha is the bits of a
    white colour set, the background is added per channel, and gradients reach colours and
    geometry."""
    from gaussianimage_amd import ops
    H, W, N = 96, 128, 1500
    means, L, col, op = synth.cholesky_inputs(N, seed=9)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    prev = ops.use_compiled(compiled)
    old = ops.get_conventions()["bg_weight"]
    try:
        with torch.no_grad():
            proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
        g = torch.Generator().manual_seed(4)
        wide = torch.rand(N, 7, generator=g).to(DEV)
        opa = op.to(DEV)
        zero = torch.zeros(3, device=DEV)

        def three(c3):
            return ops.rasterize_gaussians_sum(*proj, c3.contiguous(), opa, H, W, 16, 16,
                                               background=zero)
        for D in (1, 2, 5, 7):
            img = ops.rasterize_gaussians_sum(*proj, wide[:, :D], opa, H, W, 16, 16,
                                              background=torch.zeros(D, device=DEV))
            assert img.shape == (H, W, D)
            for ch in range(D):
                base = 3 * (ch // 3)
                c3 = torch.zeros(N, 3, device=DEV)
                k = min(3, D - base)
                c3[:, :k] = wide[:, base:base + k]
                assert torch.equal(img[..., ch], three(c3)[..., ch - base]), (D, ch)
        img, alpha = ops.rasterize_gaussians_sum(*proj, wide[:, :3].contiguous(), opa, H, W, 16, 16,
                                                 background=zero, return_alpha=True)
        assert torch.equal(img, three(wide[:, :3]))
        assert alpha.shape == (H, W)
        assert torch.equal(alpha, three(torch.ones(N, 3, device=DEV))[..., 0])
        # the background enters once per channel and never the accumulated weight
        ops.set_conventions(bg_weight=1.0)
        bg = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5], device=DEV)
        img5, alpha5 = ops.rasterize_gaussians_sum(*proj, wide[:, :5], opa, H, W, 16, 16,
                                                   background=bg, return_alpha=True)
        ops.set_conventions(bg_weight=0.0)
        plain = ops.rasterize_gaussians_sum(*proj, wide[:, :5], opa, H, W, 16, 16,
                                            background=bg)
        assert torch.allclose(img5, plain + bg, atol=1e-6) and torch.equal(alpha5, alpha)
        # gradients: five channels against the same loss written with two 3-channel calls
        c5 = wide[:, :5].clone().requires_grad_(True)
        xy = proj[0].clone().requires_grad_(True)
        con = proj[3].clone().requires_grad_(True)
        v = torch.randn(H, W, 5, generator=g).to(DEV)
        out = ops.rasterize_gaussians_sum(xy, proj[1], proj[2], con, proj[4], c5, opa, H, W, 16, 16,
                                          background=torch.zeros(5, device=DEV))
        (out * v).sum().backward()
        ca = wide[:, :3].clone().requires_grad_(True)
        cb = torch.zeros(N, 3, device=DEV)
        cb[:, :2] = wide[:, 3:5]
        cb.requires_grad_(True)
        xy2 = proj[0].clone().requires_grad_(True)
        con2 = proj[3].clone().requires_grad_(True)
        a = ops.rasterize_gaussians_sum(xy2, proj[1], proj[2], con2, proj[4], ca, opa, H, W, 16, 16,
                                        background=zero)
        b = ops.rasterize_gaussians_sum(xy2, proj[1], proj[2], con2, proj[4], cb, opa, H, W, 16, 16,
                                        background=zero)
        ((a * v[..., :3]).sum() + (b[..., :2] * v[..., 3:]).sum()).backward()
        assert torch.equal(c5.grad[:, :3], ca.grad) and torch.equal(c5.grad[:, 3:], cb.grad[:, :2])
        assert torch.allclose(xy.grad, xy2.grad, rtol=1e-5, atol=1e-7)
        assert torch.allclose(con.grad, con2.grad, rtol=1e-5, atol=1e-6)
    finally:
        ops.set_conventions(bg_weight=old)
        ops.use_compiled(prev)
