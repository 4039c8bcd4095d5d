"""The fused fitting step against the drop-in path: same parameters after K iterations as the
reference-style loop (drop-in operators + torch autograd + the reference-pinned Adan), and the
loss / image of its forward against the oracle."""
import numpy as np
import pytest
import torch

from gaussianimage_amd._testing import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _maxrel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_fused_step_matches_drop_in_training(kind):
    from gaussianimage_amd._testing import harness
    from gaussianimage_amd.fit import FusedFitter
    H, W, N, K = 96, 128, 1500, 6
    torch.manual_seed(3)
    if kind == "cholesky":
        model = harness.CholeskyImageModel(N, H, W, DEV, lr=5e-3, passes=1)
        names = ["_xyz", "_cholesky", "_features_dc"]
    else:
        model = harness.RSImageModel(N, H, W, DEV, lr=5e-3)
        names = ["_xyz", "_scaling", "_rotation", "_features_dc"]
    params = {k: getattr(model, k).detach().clone() for k in names}
    fit = FusedFitter(N, H, W, kind=kind, device=DEV, lr=5e-3, params=params)
    gt = synth.target_image(H, W).to(DEV)
    for it in range(K):
        loss, _ = model.train_iter(gt)
        fit.step(gt)
        assert abs(fit.loss() - loss.item()) <= 2e-6 * max(1.0, loss.item())
    for k in names:
        # fp32 trajectories of two implementations: equal to a few 1e-5 of the parameter range
        assert _maxrel(fit.params[k], getattr(model, k)) <= 2e-4, k
    img = model.forward()["render"]
    fit.forward(gt)
    assert float((fit.render() - img.detach()).abs().max()) <= 1e-4


def test_fit_forward_loss_and_gradient_against_oracle(oracle):
    """gi2d_fit_forward alone: image, per-tile loss sums and d mse / d image vs the oracle."""
    from gaussianimage_amd.fit import FusedFitter
    H, W, N = 80, 112, 800
    xyz, chol, feat = synth.cholesky_params(N, seed=5, footprint=2.0)
    fit = FusedFitter(N, H, W, device=DEV, params={"_xyz": xyz, "_cholesky": chol,
                                                   "_features_dc": feat})
    gt = synth.target_image(H, W).to(DEV)
    fit.forward(gt)
    means, L, col, op = synth.cholesky_inputs(N, seed=5, footprint=2.0)
    ref, margin = None, None
    xys, _, radii, conics, _ = oracle.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    offs, ids = oracle.bin_tiles(xys, radii, H, W)
    ref, margin = oracle.rasterize_sum_fwd(xys, conics, col.numpy(), op.numpy(), H, W, offs, ids,
                                           want_margin=True)
    ok = margin > 1e-3
    assert np.abs(fit.image.cpu().numpy() - ref)[ok].max() <= 2e-5
    g = gt[0].permute(1, 2, 0).cpu().numpy()
    d = np.clip(ref, 0, 1) - g
    assert abs(fit.loss() - float((d.astype(np.float64) ** 2).mean())) <= 1e-6
    v = 2 * d / d.size * ((ref >= 0) & (ref <= 1))
    assert np.abs(fit.v_image.cpu().numpy() - v)[ok].max() <= 1e-9 + 1e-4 * np.abs(v).max()


def test_fused_fit_reduces_the_loss():
    """End to end: PSNR rises over a short fit of a synthetic image."""
    from gaussianimage_amd.fit import FusedFitter
    H, W, N = 128, 192, 4000
    torch.manual_seed(0)
    fit = FusedFitter(N, H, W, device=DEV, lr=1e-2)
    gt = synth.target_image(H, W).to(DEV)
    fit.step(gt)
    first = fit.psnr()
    for _ in range(300):
        fit.step(gt)
    last = fit.psnr()
    assert last > first + 6.0, (first, last)


@pytest.mark.gpu
def test_growing_gaussians_do_not_stop_a_fit():
    """Footprints that suddenly outgrow the tile lists' capacity (sized from earlier iterations)
    cost the overflowed tiles one step's gradient, never a NaN or an exception: the workspace
    grows and the fit goes on."""
    from gaussianimage_amd._testing import synth
    from gaussianimage_amd.fit import FusedFitter
    H, W, N = 64, 80, 3000
    torch.manual_seed(2)
    fit = FusedFitter(N, H, W, kind="cholesky", device="cuda:0")
    gt = synth.target_image(H, W).to("cuda:0")
    for _ in range(5):
        fit.step(gt)
    cap0 = fit.ws.info()["capacity"]
    fit.params["_cholesky"].mul_(5.0).add_(2.0)   # every Gaussian now covers many tiles
    for _ in range(6):
        fit.step(gt)
    torch.cuda.synchronize()  # the host runs ahead of the device: let the reports arrive
    for _ in range(2):
        fit.step(gt)
    torch.cuda.synchronize()
    assert fit.overflowed_frames >= 1
    assert fit.ws.info()["capacity"] > cap0
    for p in fit.params.values():
        assert torch.isfinite(p).all()
    fit.forward(gt)
    assert torch.isfinite(fit.v_image).all() and fit.loss() == fit.loss()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N,ppl", [("cholesky", 2500, 0), ("rs", 2500, 0),
                                        ("cholesky", 150, 0), ("cholesky", 12000, 0),
                                        ("cholesky", 150, 4), ("rs", 2500, 4),
                                        ("cholesky", 6000, 4)])
def test_chained_and_one_pass_steps_are_the_plain_step(kind, N, ppl):
    """The step whose tail also projects the next iteration (3 launches), and the one that also
    runs both rasterisation passes in one kernel (2 launches), leave, bit for bit, the
    parameters, optimizer state and losses of the plain 4-launch step -- also across an outside
    modification of the parameters between two iterations, which must not be missed.  The three
    sizes give short lists (groups of lanes per entry), lists of some tens (work pieces) and
    lists beyond one staging batch; ppl = 4 forces the one-wave-per-tile kernels that sparse
    frames get (lists of a few ids, of some tens, and of more than a wave's worth)."""
    from gaussianimage_amd.fit import FusedFitter
    H, W = 96, 112
    gt = synth.target_image(H, W).to(DEV)
    fits = []
    for chain, one_pass in ((True, True), (True, False), (False, False)):
        torch.manual_seed(7)
        fits.append(FusedFitter(N, H, W, kind=kind, device=DEV, chain=chain, one_pass=one_pass))
        fits[-1].ws.set_raster_ppl(ppl)
    assert fits[0].one_pass and not fits[1].one_pass
    losses = [[], [], []]
    for it in range(9):
        for k, f in enumerate(fits):
            if it == 5:  # someone edits the parameters behind the fitter's back
                f.params["_xyz"].mul_(0.9)
            f.step(gt)
            losses[k].append(f.loss())
    assert losses[0] == losses[2] and losses[1] == losses[2]
    b = fits[2]
    for a in fits[:2]:
        for k in a.params:
            assert torch.equal(a.params[k], b.params[k]), k
            for s in a.state[k]:
                assert torch.equal(a.state[k][s], b.state[k][s]), (k, s)
    # the chained fitters' projection outputs are those of the parameters as they are now
    for f in fits:
        f.forward(gt)
    for a in fits[:2]:
        assert torch.equal(a.xys, b.xys) and torch.equal(a.radii, b.radii)
        assert torch.equal(a.conics, b.conics) and torch.equal(a.v_image, b.v_image)
        assert torch.equal(a.image, b.image)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_fit_whose_projection_fills_the_lists_through_cells_is_the_plain_fit(kind):
    """Large frames project with a launch pair of their own (lists through cells of tiles)
    instead of inside the tail: forced on a small frame, it leaves the bits of the default
    step -- parameters, optimizer state, losses, image."""
    from gaussianimage_amd import _lib
    from gaussianimage_amd.fit import FusedFitter
    N, H, W = 5000, 160, 208
    gt = synth.target_image(H, W).to(DEV)
    fits = []
    for form in (0, 4):
        torch.manual_seed(11)
        f = FusedFitter(N, H, W, kind=kind, device=DEV)
        _lib.check(_lib.lib.gi2d_workspace_set_option(f.ws.handle, _lib.OPT_PROJECT_AGG, form),
                   "set_option")
        fits.append(f)
    losses = [[], []]
    for it in range(7):
        for k, f in enumerate(fits):
            f.step(gt)
            losses[k].append(f.loss())
    assert losses[0] == losses[1]
    a, b = fits
    for k in a.params:
        assert torch.equal(a.params[k], b.params[k]), k
        for s in a.state[k]:
            assert torch.equal(a.state[k][s], b.state[k][s]), (k, s)
    for f in fits:
        f.forward(gt)
    assert torch.equal(a.image, b.image) and torch.equal(a.xys, b.xys)


@pytest.mark.gpu
def test_one_pass_step_keeps_the_image_and_survives_forward_calls_in_between():
    """gi2d_fit_step writes the optional image like the forward does, and explicit forward()
    calls between steps (which consume the lists the chained tail prepared) change nothing."""
    from gaussianimage_amd.fit import FusedFitter
    H, W, N = 70, 90, 1800   # edge tiles in both directions
    gt = synth.target_image(H, W).to(DEV)
    torch.manual_seed(3)
    a = FusedFitter(N, H, W, device=DEV)
    torch.manual_seed(3)
    b = FusedFitter(N, H, W, device=DEV, chain=False)
    for it in range(6):
        a.step(gt)
        b.step(gt)
        assert torch.equal(a.image, b.image), it
        if it % 2:
            a.forward(gt)
    for k in a.params:
        assert torch.equal(a.params[k], b.params[k]), k


@pytest.mark.gpu
def test_fused_fitter_under_the_alternative_conventions():
    """The fitting path honours the runtime conventions too (pixel centres, alpha cap, added
    background): its forward is the operator path's, bit for bit."""
    from gaussianimage_amd import ops
    from gaussianimage_amd.fit import FusedFitter
    before = ops.get_conventions()
    ops.set_conventions(pixel_offset=0.5, center_bias=-0.5, alpha_cap=0.999, bg_weight=0.25)
    try:
        H, W, N = 80, 96, 1200
        torch.manual_seed(4)
        fit = FusedFitter(N, H, W, kind="cholesky", device=DEV)
        gt = synth.target_image(H, W).to(DEV)
        fit.forward(gt)
        tb = ((W + 15) // 16, (H + 15) // 16, 1)
        proj = ops.project_gaussians_2d(torch.tanh(fit.params["_xyz"]),
                                        fit.params["_cholesky"] + torch.tensor(
                                            [0.5, 0.0, 0.5], device=DEV), H, W, tb)
        img = ops.rasterize_gaussians_sum(*proj, fit.params["_features_dc"], fit._opacity, H, W,
                                          16, 16, background=fit.background)
        assert torch.equal(fit.image, img)
        assert float(img.min()) >= 0.25  # the background is in
        fit.step(gt)
        fit.step(gt)
        assert all(torch.isfinite(p).all() for p in fit.params.values())
    finally:
        ops.set_conventions(**before)


@pytest.mark.gpu
def test_spatial_order_is_a_permutation_of_the_same_fit():
    """spatial_order=True stores the Gaussians in tile order (shared list atomics): the same
    model up to a permutation, so the fit follows the same losses up to summation order."""
    from gaussianimage_amd.fit import FusedFitter
    H, W, N = 128, 160, 4000
    gt = synth.target_image(H, W).to(DEV)
    torch.manual_seed(5)
    a = FusedFitter(N, H, W, device=DEV)
    torch.manual_seed(5)
    b = FusedFitter(N, H, W, device=DEV, spatial_order=True)
    for k in a.params:   # same multiset of rows
        assert torch.equal(torch.sort(a.params[k].sum(1)).values,
                           torch.sort(b.params[k].sum(1)).values), k
    m = torch.tanh(b.params["_xyz"])
    tile = ((0.5 * H * m[:, 1] + 0.5 * H) / 16).floor() * 1000 + ((0.5 * W * m[:, 0] + 0.5 * W) / 16).floor()
    assert bool((tile[1:] >= tile[:-1]).all())
    for it in range(12):
        a.step(gt)
        b.step(gt)
        la, lb = a.loss(), b.loss()
        assert abs(la - lb) <= 1e-5 * la, (it, la, lb)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(0, 40, 56), (1, 16, 16), (5, 1, 1), (300, 33, 17)])
def test_fitter_on_degenerate_sizes(N, H, W):
    """No Gaussians, one Gaussian, a one-pixel image, odd sizes: the fused step runs, the loss is
    the plain step's, nothing is NaN."""
    from gaussianimage_amd.fit import FusedFitter
    gt = synth.target_image(H, W).to(DEV)
    torch.manual_seed(9)
    a = FusedFitter(N, H, W, device=DEV)
    torch.manual_seed(9)
    b = FusedFitter(N, H, W, device=DEV, chain=False)
    for _ in range(4):
        a.step(gt)
        b.step(gt)
        assert a.loss() == b.loss() and a.loss() == a.loss()
    for k in a.params:
        assert torch.equal(a.params[k], b.params[k]) and torch.isfinite(a.params[k]).all()
    if N == 0:
        want = float((gt.clamp(0, 1) ** 2).mean())   # an empty model renders black
        assert abs(a.loss() - want) <= 1e-6 * max(want, 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_split_step_with_one_band_is_the_fused_step(kind):
    """gi2d_fit_raster + gi2d_fit_tail over the whole frame (the tile-row sharded step with one
    rank, no exchange) leave the bits of gi2d_fit_step."""
    from gaussianimage_amd.fit import FusedFitter
    from gaussianimage_amd.sharding import ShardedFitter
    H, W, N = 96, 112, 2500
    gt = synth.target_image(H, W).to(DEV)
    torch.manual_seed(11)
    a = FusedFitter(N, H, W, kind=kind, device=DEV, keep_image=False)
    torch.manual_seed(11)
    b = ShardedFitter(N, H, W, kind, 0, 1, DEV)
    for it in range(7):
        a.step(gt)
        b.step(gt)
        assert a.loss() == b.loss(), it
    for k in a.params:
        assert torch.equal(a.params[k], b.params[k]), k
        for s in a.state[k]:
            assert torch.equal(a.state[k][s], b.fit.state[k][s]), (k, s)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", [("cholesky", 2), ("cholesky", 3), ("rs", 4)])
def test_sharded_fit_on_one_device_follows_the_unsharded_fit(kind, world):
    """Tile-row sharded fitting emulated on one device: `world` fitters, each with its band,
    exchange their per-Gaussian partial gradients by a plain sum (what the all-reduce does) and
    apply the same update.  The replicas stay bit-identical to each other, and follow the
    unsharded fused fit up to the re-association of the per-band sums."""
    from gaussianimage_amd.fit import FusedFitter
    from gaussianimage_amd.sharding import ShardedFitter
    H, W, N = 120, 112, 3000     # 8 tile rows: uneven bands for world = 3
    gt = synth.target_image(H, W).to(DEV)
    torch.manual_seed(12)
    ref = FusedFitter(N, H, W, kind=kind, device=DEV, keep_image=False)
    ranks = []
    for r in range(world):
        torch.manual_seed(12)
        ranks.append(ShardedFitter(N, H, W, kind, r, world, DEV))
    assert sorted(x for f in ranks for x in range(*f.rows)) == list(range((H + 15) // 16))

    class SumOverRanks:      # stands in for the process group's all-reduce
        def __init__(self):
            self.pending = []

        def reduce_all(self):
            total = torch.stack([f.exchange.flat for f in ranks]).sum(0)
            for f in ranks:
                f.exchange.flat.copy_(total)

    from gaussianimage_amd import _lib
    from gaussianimage_amd.ops import _ptr
    import ctypes
    for it in range(8):
        ref.step(gt)
        # the three phases of ShardedFitter.step, rank by rank, with the sum in between
        for f in ranks:
            f.fit._check_target(gt)
            rc = _lib.lib.gi2d_fit_raster(*f._raster_args, _ptr(gt), *f._raster_tail, f.fit._stream)
            _lib.check(f.fit._late_overflow(rc), "gi2d_fit_raster")
        SumOverRanks().reduce_all()
        for f in ranks:
            ff = f.fit
            lr = ff.lr
            ff.step_count += 1
            b1, b2, b3 = ff.betas
            hyper = _lib.AdanHyper(lr, b1, b2, b3, ff.eps, ff.weight_decay, 0, ff.step_count)
            _lib.check(_lib.lib.gi2d_fit_tail(*f._tail_args, ctypes.byref(hyper), ff._stream),
                       "gi2d_fit_tail")
            ff._versions = ff._param_versions()
        loss = sum(float(f.fit.loss_partials[: f.band_tiles].double().sum()) for f in ranks) / (
            3 * H * W)
        assert abs(loss - ref.loss()) <= 1e-6 * max(ref.loss(), 1e-12), it
    for k in ref.params:
        for f in ranks[1:]:
            assert torch.equal(f.params[k], ranks[0].params[k]), k
        assert _maxrel(ranks[0].params[k], ref.params[k]) <= 2e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("kind,F,N,H,W,K,form", [("cholesky", 3, 1500, 96, 128, 25, 0),
                                                 ("rs", 2, 1200, 80, 112, 25, 0),
                                                 ("cholesky", 5, 30000, 512, 768, 12, 0),
                                                 ("rs", 4, 2500, 160, 208, 12, 4),
                                                 ("cholesky", 2, 3000, 144, 240, 12, 4)])
def test_batch_fitter_gives_every_image_its_own_fit(kind, F, N, H, W, K, form):
    """F images fitted in lockstep by one call of two launches: image f's parameters, optimizer
    state and losses are, bit for bit, those of a FusedFitter that fits it alone -- also across
    an outside edit of the parameters (the chained projection is dropped and redone).  form 4:
    the batch fills its lists through cells of tiles (as batches of large frames do by
    themselves), the single fits do not."""
    from gaussianimage_amd import _lib
    from gaussianimage_amd.fit import BatchFitter, FusedFitter
    per = []
    for f in range(F):
        if kind == "cholesky":
            xyz, chol, feat = synth.cholesky_params(N, seed=40 + f)
            per.append({"_xyz": xyz, "_cholesky": chol, "_features_dc": feat})
        else:
            xyz, sc, rot, feat = synth.rs_params(N, seed=40 + f)
            per.append({"_xyz": xyz, "_scaling": sc, "_rotation": rot, "_features_dc": feat})
    gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(F)]).to(DEV).contiguous()
    singles = [FusedFitter(N, H, W, kind=kind, device=DEV, lr=5e-3, params=p, keep_image=False)
               for p in per]
    batch = BatchFitter(F, N, H, W, kind=kind, device=DEV, lr=5e-3,
                        params={k: torch.stack([p[k] for p in per]) for k in per[0]})
    _lib.check(_lib.lib.gi2d_workspace_set_option(batch.ws.handle, _lib.OPT_PROJECT_AGG, form),
               "set_option")
    for it in range(K):
        if it == K // 2:
            # an edit from outside between two iterations
            for fit in singles:
                fit.params["_features_dc"].mul_(0.9)
            batch.params["_features_dc"].mul_(0.9)
        for f, fit in enumerate(singles):
            fit.step(gts[f:f + 1])
        batch.step(gts)
        if it % 6 == 5 or it == K - 1:
            losses = batch.losses()
            for f, fit in enumerate(singles):
                # (the per-tile sums are the kernels'; their total is added up on the host)
                assert torch.equal(batch.loss_partials[f], fit.loss_partials), (it, f)
                assert abs(float(losses[f]) - fit.loss()) <= 1e-12
    assert batch.overflowed_frames == 0
    for f, fit in enumerate(singles):
        for k in fit.params:
            assert torch.equal(batch.params[k][f], fit.params[k]), (f, k)
            for s in fit.state[k]:
                assert torch.equal(batch.state[k][s][f], fit.state[k][s]), (f, k, s)
    assert float(batch.losses()[0]) < 1.0


@pytest.mark.gpu
def test_batch_of_24_at_full_size_is_the_single_fit_of_the_same_workgroup_shape():
    """24 images of 768x512 / 30 000 Gaussians in one call: the batch takes one wave per tile
    (36 864 tiles in the launch); single fits forced to that shape leave the same bits.  (Left
    to its own four waves per tile a single fit adds a Gaussian's per-pixel gradient terms in
    another order and agrees only to rounding.)"""
    from gaussianimage_amd.fit import BatchFitter, FusedFitter
    F, N, H, W = 24, 30000, 512, 768
    per = []
    for f in range(F):
        xyz, chol, feat = synth.cholesky_params(N, seed=40 + f)
        per.append({"_xyz": xyz, "_cholesky": chol, "_features_dc": feat})
    gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(F)]).to(DEV).contiguous()
    batch = BatchFitter(F, N, H, W, kind="cholesky", device=DEV, lr=5e-3,
                        params={k: torch.stack([p[k] for p in per]) for k in per[0]})
    picks = (0, 11, 23)
    singles = {f: FusedFitter(N, H, W, kind="cholesky", device=DEV, lr=5e-3, params=per[f],
                              keep_image=False) for f in picks}
    for s in singles.values():
        s.ws.set_raster_ppl(4)
    for it in range(5):
        for f, s in singles.items():
            s.step(gts[f:f + 1])
        batch.step(gts)
    for f, s in singles.items():
        assert torch.equal(batch.loss_partials[f], s.loss_partials), f
        for k in s.params:
            assert torch.equal(batch.params[k][f], s.params[k]), (f, k)


@pytest.mark.gpu
@pytest.mark.parametrize("F", [2, 5])
def test_growing_gaussians_do_not_stop_a_batch_fit(F):
    """The batch fit under the same stress as the single fit: footprints that outgrow the lists'
    capacity cost the overflowed tiles one step's gradient, the workspace grows, nothing becomes
    NaN -- with the chained tail (fewer than four images) and with the separate batch projection."""
    from gaussianimage_amd.fit import BatchFitter
    H, W, N = 64, 80, 3000
    torch.manual_seed(2)
    fit = BatchFitter(F, N, H, W, kind="cholesky", device=DEV)
    gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(F)]).to(DEV).contiguous()
    for _ in range(5):
        fit.step(gts)
    cap0 = fit.ws.info()["capacity"]
    fit.params["_cholesky"].mul_(5.0).add_(2.0)
    for _ in range(6):
        fit.step(gts)
    torch.cuda.synchronize()
    for _ in range(2):
        fit.step(gts)
    torch.cuda.synchronize()
    assert fit.overflowed_frames >= 1
    assert fit.ws.info()["capacity"] > cap0
    for p in fit.params.values():
        assert torch.isfinite(p).all()
    for _ in range(3):
        fit.step(gts)
    assert torch.isfinite(fit.losses()).all()


@pytest.mark.gpu
def test_batch_fitter_rejects_what_it_cannot_fit():
    from gaussianimage_amd.fit import BatchFitter
    with pytest.raises(ValueError):
        BatchFitter(0, 10, 32, 32, device=DEV)
    with pytest.raises(ValueError):
        BatchFitter(2, 10, 32, 32, device=DEV, params={"_xyz": torch.zeros(10, 2),
                                                        "_cholesky": torch.zeros(10, 3),
                                                        "_features_dc": torch.zeros(10, 3)})
    fit = BatchFitter(2, 10, 32, 32, device=DEV)
    with pytest.raises(ValueError):
        fit.step(torch.zeros(1, 3, 32, 32, device=DEV))
    fit.step(torch.zeros(2, 3, 32, 32, device=DEV))   # (and an empty-ish problem just runs)
    assert torch.isfinite(fit.losses()).all()


@pytest.mark.gpu
@pytest.mark.parametrize("F,N,H,W", [(2, 0, 40, 56), (5, 1, 16, 16), (3, 5, 1, 1), (4, 300, 33, 17)])
def test_batch_fitter_on_degenerate_sizes(F, N, H, W):
    """The same degenerate problems in a batch: every image follows its single fit."""
    from gaussianimage_amd.fit import BatchFitter, FusedFitter
    gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(F)]).to(DEV).contiguous()
    torch.manual_seed(9)
    batch = BatchFitter(F, N, H, W, device=DEV)
    singles = [FusedFitter(N, H, W, device=DEV,
                           params={k: v[f] for k, v in batch.params.items()}) for f in range(F)]
    for _ in range(4):
        batch.step(gts)
        for f, fit in enumerate(singles):
            fit.step(gts[f:f + 1])
    for f, fit in enumerate(singles):
        assert torch.equal(batch.loss_partials[f], fit.loss_partials), f
        for k in fit.params:
            assert torch.equal(batch.params[k][f], fit.params[k]), (f, k)
            assert torch.isfinite(batch.params[k]).all()
