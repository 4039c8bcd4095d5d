"""The batched frame call and the rasteriser's per-Gaussian records.

* gi2d_render_batch: every frame of a batch is bit-identical to the frame gi2d_render_frame
  renders from the same inputs, projection outputs and tile lists included -- on the first call
  of a shape (lists sized synchronously) and in steady state, for both covariance forms, odd
  sizes and a batch of one;
* the records (one packed 48-byte line per Gaussian, written by the projections that know colours
  and opacities): switched off, every path that uses them -- frame call, batch, codec decode,
  fused fitting -- gives the same bits."""
import numpy as np
import pytest
import torch

from gaussianimage_amd._testing import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(kind, N, seed, fp):
    if kind == "cholesky":
        means, a, col, op = synth.cholesky_inputs(N, seed=seed, footprint=fp)
        b = None
    else:
        means, a, b, col, op = synth.rs_inputs(N, seed=seed, footprint=fp)
    g = torch.Generator().manual_seed(seed + 50)
    op = 0.05 + 0.95 * torch.rand(N, generator=g)   # varied opacities: plain entries with lo != 0
    return means, a, b, col, op


@pytest.mark.parametrize("kind,F,N,H,W,fp", [("cholesky", 5, 3000, 96, 128, 2.0),
                                             ("rs", 3, 1500, 70, 90, 1.0),
                                             ("cholesky", 1, 800, 48, 64, 1.0),
                                             ("cholesky", 4, 30000, 512, 768, 1.0),
                                             ("rs", 2, 40, 33, 17, 1.0)])
def test_batch_equals_single_frames(kind, F, N, H, W, fp):
    from gaussianimage_amd.pipeline import FrameBatch, FramePipeline
    per = [_inputs(kind, N, 10 + f, fp) for f in range(F)]
    stack = [None if per[0][k] is None else
             torch.stack([p[k].reshape(N, -1) for p in per]).to(DEV).contiguous() for k in range(5)]
    stack[4] = stack[4].reshape(F, N)
    batch = FrameBatch(F, N, H, W, kind=kind, device=DEV)
    single = FramePipeline(N, H, W, kind=kind, device=DEV)
    for rep in range(3):   # the first call learns the list capacity synchronously
        imgs = batch.forward(*stack).clone()
        offs, ids = batch.ws.bins()
        T = single.tiles_x * single.tiles_y
        assert offs.numel() == F * T + 1
        for f in range(F):
            args = [None if t is None else t[f].contiguous() for t in stack]
            img = single.forward(*args)
            assert torch.equal(imgs[f], img), (rep, f)
            for name in ("xys", "depths", "radii", "conics", "num_tiles_hit"):
                assert torch.equal(getattr(batch, name)[f], getattr(single, name)), (rep, f, name)
            if rep == 0:
                # (a shape's first call bins every tile of the radius boxes, later calls only
                # the tiles the supports can touch: compare lists made the same way)
                continue
            s_offs, s_ids = single.ws.bins()
            b_offs = offs[f * T:(f + 1) * T + 1]
            assert torch.equal(b_offs - b_offs[0], s_offs), (rep, f)
            assert torch.equal(ids[int(b_offs[0]):int(b_offs[-1])], s_ids), (rep, f)
    assert batch.ws.info()["frames"] == F and batch.ws.info()["records"]


def test_the_benched_batch_against_the_oracle(oracle):
    """The shape bench.py times -- 24 frames of 768x512 / 30 000 Gaussians in one call, every
    form left to the library's own choice (LDS-aggregated projection, one wave per tile, packed
    records, culled lists) -- compared with the oracle directly, not through other entry points:
    projection outputs bit for bit, images within 1e-5 (pixels on the alpha cut set aside), and,
    with the support culling switched off, the tile lists bit for bit."""
    import bench
    from gaussianimage_amd.pipeline import FrameBatch
    F, (H, W, N, kind, fp) = 24, bench.WORKLOADS["kodak30k"]
    stack = bench.make_batch_inputs("kodak30k", DEV, F, seed=1)
    batch = FrameBatch(F, N, H, W, kind=kind, device=DEV)
    run = batch.bind(*stack)
    for _ in range(3):
        run()
    imgs = batch.images.clone()
    T = bench.tiles_of(H, W)
    offs, ids = batch.ws.bins()
    batch.ws.set_cull(False)
    run()
    assert torch.equal(batch.images, imgs)       # (culling drops only pairs that contribute nothing)
    full_offs, full_ids = batch.ws.bins()
    batch.ws.set_cull(True)
    assert int(offs[-1]) < int(full_offs[-1])
    for f in (0, 13, 23):
        means, L, col, op = (stack[0][f].cpu().numpy(), stack[1][f].cpu().numpy(),
                             stack[3][f].cpu().numpy(), stack[4][f].cpu().numpy().reshape(N, 1))
        r_xys, r_depths, r_radii, r_conics, r_nth = oracle.project_cholesky_fwd(means, L, H, W)
        assert np.array_equal(batch.xys[f].cpu().numpy(), r_xys), f
        assert np.array_equal(batch.radii[f].cpu().numpy(), r_radii), f
        assert np.array_equal(batch.conics[f].cpu().numpy(), r_conics), f
        assert np.array_equal(batch.num_tiles_hit[f].cpu().numpy(), r_nth), f
        r_offs, r_ids = oracle.bin_tiles(r_xys, r_radii, H, W)
        b_offs = full_offs[f * T:(f + 1) * T + 1].numpy()
        assert np.array_equal(b_offs - b_offs[0], r_offs), f
        assert np.array_equal(full_ids[int(b_offs[0]):int(b_offs[-1])].numpy(), r_ids), f
        ref, margin = oracle.rasterize_sum_fwd(r_xys, r_conics, col, op, H, W, r_offs, r_ids,
                                               want_margin=True)
        ok = margin > 1e-4
        assert ok.mean() > 0.995
        err = np.abs(imgs[f].cpu().numpy() - ref)
        assert err[ok].max() <= 1e-5, (f, err[ok].max())
        assert err.max() <= 3.0 / 255 * max(1.0, float(col.max()))


def test_batch_argument_errors():
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FrameBatch
    F, N, H, W = 2, 100, 32, 32
    batch = FrameBatch(F, N, H, W, device=DEV)
    means, L, _, col, op = _inputs("cholesky", N, 1, 1.0)
    ok = [means.expand(F, N, 2).contiguous().to(DEV), L.expand(F, N, 3).contiguous().to(DEV), None,
          col.expand(F, N, 3).contiguous().to(DEV), op.expand(F, N).contiguous().to(DEV)]
    batch.forward(*ok)
    with pytest.raises(ValueError):
        batch.forward(ok[0][:1], *ok[1:])
    with pytest.raises(ValueError):
        FrameBatch(0, N, H, W, device=DEV)
    # a band and a batch do not combine; the single-frame calls refuse a batch's lists
    batch.ws.set_rows(0, 1)
    with pytest.raises(_lib.Gi2dError):
        batch.forward(*ok)
    batch.ws.set_rows()
    batch.forward(*ok)
    import ctypes
    z = torch.zeros(N, 3, device=DEV)
    rc = _lib.lib.gi2d_rasterize_sum_fwd(batch.ws.handle, batch.xys.data_ptr(),
                                         batch.conics.data_ptr(), z.data_ptr(),
                                         ok[4].data_ptr(), None, N, H, W,
                                         ctypes.byref(batch.conv), batch.images.data_ptr(), None)
    assert rc == _lib.GI2D_E_STATE


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_records_change_no_bit_of_the_frame_call(kind):
    from gaussianimage_amd.pipeline import FrameBatch, FramePipeline
    N, H, W = 6000, 120, 200
    ins = _inputs(kind, N, 3, 2.5)
    ins[4][::7] = 0.002                      # below the cut: listed nowhere
    ins[4][1::11] = 1.5                      # above the cap: general entries
    args = [None if t is None else t.to(DEV).contiguous() for t in ins]
    imgs = []
    for records in (True, False):
        pipe = FramePipeline(N, H, W, kind=kind, device=DEV)
        pipe.ws.set_records(records)
        for _ in range(2):
            img = pipe.forward(*args).clone()
        assert pipe.ws.info()["records"] == records
        imgs.append(img)
    assert torch.equal(imgs[0], imgs[1])
    # ... and the operator path (separate tensors, never any records) gives the same image
    from gaussianimage_amd import ops
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    if kind == "cholesky":
        proj = ops.project_gaussians_2d(args[0], args[1], H, W, tb)
    else:
        proj = ops.project_gaussians_2d_scale_rot(args[0], args[1], args[2], H, W, tb)
    img = ops.rasterize_gaussians_sum(*proj, args[3], args[4].reshape(N, 1), H, W, 16, 16)
    assert torch.equal(img, imgs[0])


@pytest.mark.parametrize("kind", ["cholesky", "rs"])
def test_records_change_no_bit_of_a_fit_or_a_decode(kind):
    from gaussianimage_amd import codec
    from gaussianimage_amd.fit import FusedFitter
    N, H, W = 2500, 96, 112
    gt = synth.target_image(H, W).to(DEV)
    fits = []
    for records in (True, False):
        torch.manual_seed(7)
        f = FusedFitter(N, H, W, kind=kind, device=DEV)
        f.ws.set_records(records)
        for _ in range(8):
            f.step(gt)
        fits.append(f)
        assert f.ws.info()["records"] == records
    for k in fits[0].params:
        assert torch.equal(fits[0].params[k], fits[1].params[k]), k
    assert fits[0].loss() == fits[1].loss() and torch.equal(fits[0].image, fits[1].image)
    enc, q = codec.synthetic_encoding(N, kind, seed=2)
    outs = []
    for records in (True, False):
        dec = codec.CodecDecoder(enc, q, H, W, kind=kind, device=DEV)
        dec.ws.set_records(records)
        for _ in range(2):
            out = dec.decode()["render"].clone()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kind,F,N,H,W,fp", [("cholesky", 1, 20000, 256, 320, 1.0),
                                             ("rs", 2, 9000, 96, 128, 2.0),
                                             ("cholesky", 6, 1500, 33, 17, 1.0),
                                             ("cholesky", 1, 70000, 512, 768, 1.0),
                                             ("rs", 5, 30000, 512, 768, 1.0),
                                             ("cholesky", 3, 4000, 700, 40, 3.0),
                                             ("cholesky", 1, 8192, 512, 512, 20.0),
                                             ("rs", 2, 3000, 256, 256, 6.0),
                                             ("cholesky", 1, 40000, 1000, 1500, 1.0),
                                             ("rs", 1, 5000, 300, 200, 20.0),
                                             ("cholesky", 1, 2500, 2100, 40, 2.0),
                                             ("cholesky", 1, 2048, 512, 512, 60.0),
                                             # (odd tile counts per frame: the counters pair up
                                             # inside a frame)
                                             ("cholesky", 5, 6000, 200, 176, 1.0),
                                             ("rs", 3, 2500, 72, 112, 2.0)])
def test_list_filling_forms_give_the_same_lists(kind, F, N, H, W, fp):
    """The projection that counts its tiles in LDS (one counter atomic per tile a workgroup
    touches) and the one that fills the lists by bands of tile rows (no counter atomics) against
    the one-atomic-per-pair form: same outputs, same lists, same images -- forced on for single
    frames, for runs of several Gaussians per lane (70 000 Gaussians: the workgroups-per-frame
    floor of the first, beyond the limit of the second), for a frame smaller than one workgroup
    for a tall frame whose bands hold several tile rows, and for footprints so large that a
    workgroup's (tile, Gaussian) pairs exceed what it can gather in LDS (8192 Gaussians, ~20
    tiles each) or nearly do.  Form 4 (through cells of tiles, single frames): several hundred
    cells, footprints over many cells (more pairs than either pass gathers in LDS), a frame one
    cell wide."""
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FrameBatch
    per = [_inputs(kind, N, 30 + f, fp) for f in range(F)]
    stack = [None if per[0][k] is None else
             torch.stack([p[k].reshape(N, -1) for p in per]).to(DEV).contiguous() for k in range(5)]
    stack[4] = stack[4].reshape(F, N)
    got = []
    for mode in (2, 1, 3, 4):
        b = FrameBatch(F, N, H, W, kind=kind, device=DEV)
        _lib.check(_lib.lib.gi2d_workspace_set_option(b.ws.handle, _lib.OPT_PROJECT_AGG, mode),
                   "set_option")
        for _ in range(3):
            imgs = b.forward(*stack).clone()
        offs, ids = b.ws.bins()
        got.append((imgs, offs, ids, b.xys.clone(), b.radii.clone(), b.conics.clone(),
                    b.num_tiles_hit.clone()))
    for other in got[1:]:
        for x, y in zip(got[0], other):
            assert torch.equal(x, y)
    if fp == 20.0:   # (8 workgroups of 1024 Gaussians, 16384 pairs gathered at most)
        assert int(got[0][1][-1]) > 8 * 16384
    if fp == 60.0:   # (2 workgroups, hundreds of tiles per Gaussian)
        assert int(got[0][1][-1]) > 2 * 16 * 16384


@pytest.mark.parametrize("form", [4, 3])
def test_a_failed_side_buffer_allocation_falls_back_to_another_form(form):
    """The cells form needs buckets, the bands form a box per Gaussian.  When that allocation
    fails the projection takes a form that needs none -- no error, no sticky HIP error, the same
    outputs, lists and images -- and takes the requested form again once memory is there."""
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FrameBatch
    kind, F, N, H, W = "cholesky", 2, 9000, 256, 320
    per = [_inputs(kind, N, 40 + f, 1.5) for f in range(F)]
    stack = [None if per[0][k] is None else
             torch.stack([p[k].reshape(N, -1) for p in per]).to(DEV).contiguous() for k in range(5)]
    stack[4] = stack[4].reshape(F, N)

    def state(b):
        offs, ids = b.ws.bins()
        return (b.images.clone(), offs, ids, b.xys.clone(), b.radii.clone(), b.num_tiles_hit.clone())

    ref = FrameBatch(F, N, H, W, kind=kind, device=DEV)
    _lib.check(_lib.lib.gi2d_workspace_set_option(ref.ws.handle, _lib.OPT_PROJECT_AGG, 2), "opt")
    for _ in range(3):
        ref.forward(*stack)
    want = state(ref)
    b = FrameBatch(F, N, H, W, kind=kind, device=DEV)
    _lib.check(_lib.lib.gi2d_workspace_set_option(b.ws.handle, _lib.OPT_PROJECT_AGG, form), "opt")
    for failing in (1, 0, 1, 0):
        _lib.check(_lib.lib.gi2d_workspace_set_option(b.ws.handle, _lib.OPT_FAIL_SIDE_ALLOCS,
                                                      failing), "opt")
        for _ in range(3):
            b.forward(*stack)
        for x, y in zip(state(b), want):
            assert torch.equal(x, y), failing
        torch.cuda.synchronize()     # (raises if a HIP error was left behind)


def test_cells_form_on_random_shapes_and_bands():
    """List filling through cells of tiles against one counter atomic per pair on a seeded
    sweep of frame sizes (cells cut by the frame's edges), Gaussian counts (less than one
    workgroup to several), footprints and bands of tile rows that start and end inside cells:
    same outputs, same lists, same pixels."""
    import random
    from gaussianimage_amd import _lib
    from gaussianimage_amd.pipeline import FramePipeline
    rnd = random.Random(5)
    for case in range(14):
        kind = rnd.choice(["cholesky", "rs"])
        H, W = rnd.randint(17, 700), rnd.randint(17, 700)
        N = rnd.choice([1, 37, 900, 1024, 1025, 3000, 6000])
        fp = rnd.choice([0.5, 1.0, 3.0, 10.0])
        tiles_y = (H + 15) // 16
        r0 = rnd.randint(0, tiles_y - 1)
        rows = None if case % 3 == 0 else (r0, rnd.randint(r0 + 1, tiles_y))
        inputs = [None if t is None else t.to(DEV).contiguous() for t in _inputs(kind, N, case, fp)]
        inputs[4] = inputs[4].reshape(-1)
        got = []
        for form in (2, 4):
            pipe = FramePipeline(N, H, W, kind=kind, device=DEV, rows=rows)
            _lib.check(_lib.lib.gi2d_workspace_set_option(pipe.ws.handle, _lib.OPT_PROJECT_AGG,
                                                          form), "set_option")
            for _ in range(3):
                img = pipe.forward(*inputs).clone()
            offs, ids = pipe.ws.bins()
            got.append((img, offs, ids, pipe.xys.clone(), pipe.radii.clone(),
                        pipe.conics.clone(), pipe.num_tiles_hit.clone()))
        for x, y in zip(*got):
            assert torch.equal(x, y), (case, kind, H, W, N, fp, rows)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_batch_ring_gives_the_bits_of_the_batch_call(depth):
    """Batches in flight on two streams (the projection of one under the rasteriser of the one
    before): every image is the one-call batch's, whatever the depth and whichever slot."""
    from gaussianimage_amd.pipeline import BatchRing, FrameBatch
    kind, F, N, H, W = "cholesky", 4, 6000, 160, 208
    batches = []
    for s in range(3):
        per = [_inputs(kind, N, 70 + 10 * s + f, 1.0) for f in range(F)]
        stack = [None if per[0][k] is None else
                 torch.stack([p[k].reshape(N, -1) for p in per]).to(DEV).contiguous()
                 for k in range(5)]
        stack[4] = stack[4].reshape(F, N)
        batches.append(stack)
    ref = FrameBatch(F, N, H, W, kind=kind, device=DEV)
    want = [ref.forward(*b).clone() for b in batches]
    torch.cuda.synchronize()
    ring = BatchRing(F, N, H, W, kind=kind, device=DEV, depth=depth)
    for rep in range(7):
        b = rep % 3
        imgs, done = ring.render(*batches[b])
        done.synchronize()
        assert torch.equal(imgs, want[b]), (depth, rep)
    # and without waiting in between: the slots' buffers are only reused `depth` calls later
    outs = []
    for rep in range(depth):
        imgs, done = ring.render(*batches[rep % 3])
        outs.append((imgs, done, rep % 3))
    for imgs, done, b in outs:
        done.synchronize()
        assert torch.equal(imgs, want[b])
    ring.synchronize()
