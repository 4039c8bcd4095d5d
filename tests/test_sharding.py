"""Tile-row sharding on CPU: the band plan, and the exchange step (all-gather of bands,
all-reduce of partial gradients) with two gloo ranks.  Band contents are produced by the CPU
oracle (rendering itself is GPU-only); what is tested is that the partition plus the exchange
reproduce the unsharded frame and gradients."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussianimage_amd import sharding
from gaussianimage_amd._testing import synth
from gaussianimage_amd.sharding import band_pixels, band_plan


def test_band_plan_covers_every_row_once():
    for tiles_y in (0, 1, 5, 32, 48, 135):
        for world in (1, 2, 3, 4, 8):
            plan = band_plan(tiles_y, world)
            assert len(plan) == world
            rows = [r for b, e in plan for r in range(b, e)]
            assert rows == list(range(tiles_y))
            sizes = [e - b for b, e in plan]
            # equal slots: everything before the first short band has the maximum size
            rm = max(sizes) if sizes else 0
            short = [i for i, s in enumerate(sizes) if s < rm]
            if short:
                assert all(s == rm for s in sizes[:short[0]])
                assert all(s == 0 for s in sizes[short[0] + 1:])
    assert band_plan(135, 8) == [(0, 17), (17, 34), (34, 51), (51, 68), (68, 85), (85, 102),
                                 (102, 119), (119, 135)]
    assert band_pixels(band_plan(135, 8), 7, 2160) == (1904, 2160)


def _oracle_frame(H, W, N, seed):
    from oracle import cport
    means, L, col, op = synth.cholesky_inputs(N, seed=seed, footprint=1.5)
    xys, _, radii, conics, _ = cport.project_cholesky_fwd(means.numpy(), L.numpy(), H, W)
    offs, ids = cport.bin_tiles(xys, radii, H, W)
    img = cport.rasterize_sum_fwd(xys, conics, col.numpy(), op.numpy(), H, W, offs, ids)
    return (xys, radii, conics, col.numpy(), op.numpy(), offs, ids), img


def _worker(rank, world, path, H, W, N, out):
    from oracle import cport
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        (xys, radii, conics, col, op, offs, ids), img = _oracle_frame(H, W, N, seed=3)
        g = sharding.BandGather(H, W, world, rank, "cpu")
        y0, y1 = band_pixels(g.plan, rank, H)
        for turn in range(3):  # slots rotate; every turn must reassemble the frame
            g.band_view().copy_(torch.from_numpy(img[y0:y1]) * (turn + 1))
            full, _ = g.gather()
            assert full.shape == (H, W, 3)
            assert np.array_equal(full.numpy(), img * (turn + 1))
        for turn in range(2):  # the same frame collected on one rank only
            g.band_view().copy_(torch.from_numpy(img[y0:y1]) * (turn + 2))
            got = g.gather_to(dst=1)
            if rank == 1:
                assert np.array_equal(got.numpy(), img * (turn + 2))
            else:
                assert got is None
        # backward: a band's partial gradients come from v_out restricted to the band
        v_out = np.random.default_rng(5).standard_normal((H, W, 3)).astype(np.float32)
        v_band = np.zeros_like(v_out)
        v_band[y0:y1] = v_out[y0:y1]
        part = cport.rasterize_sum_bwd(xys, conics, col, op, H, W, offs, ids, v_band)
        grads = [torch.from_numpy(p.copy()) for p in part]
        sharding.all_reduce_grads(grads)
        ref = cport.rasterize_sum_bwd(xys, conics, col, op, H, W, offs, ids, v_out)
        for a, b in zip(grads, ref):
            assert np.abs(a.numpy() - b).max() <= 1e-5 * max(1.0, np.abs(b).max())
        out.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(96, 80), (72, 64)])
def test_two_rank_gather_and_grad_reduce_gloo(H, W):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "rdzv")
        procs = [ctx.Process(target=_worker, args=(r, world, path, H, W, 300, out))
                 for r in range(world)]
        for p in procs:
            p.start()
        res = [out.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
