#!/usr/bin/env python3
"""Benchmark of the hot path: project -> tile binning -> rasterize_sum (forward), through the
C ABI, on synthetic parameters with the reference models' own initial distributions.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" is one forward pass of the hot path over one frame whose operator inputs are
already resident in HBM.  N=1 runs BASELINE config 2 (768x512, 30k Gaussians, Cholesky).
For N>1 (launched under torch.distributed.run, one rank per GPU) every rank renders its own
frames of that configuration -- frames are independent units, so there is no data-path
collective and scaling is weak -- and, as a secondary measurement reported under
``tile_row_sharded``, the ranks render ONE 3840x2160 / 200k frame split by tile rows with an
RCCL all-gather of the bands.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      -- the dominant kernel (rasterize_sum forward) against the HBM roofline,
  cpu_baseline  -- this repository's CPU oracle timed on the host cores (a reported
                   baseline; the reference has no CPU path),
  extras        -- forward+backward ms, per-stage times, the reference-style FPS loop.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (H, W, N, kind, footprint)
    "kodak30k": (512, 768, 30000, "cholesky", 1.0),        # BASELINE config 2 (headline)
    "kodak30k_rs": (512, 768, 30000, "rs", 1.0),           # config 3
    "kodak5k": (512, 768, 5000, "cholesky", 1.0),          # config 1
    "div2k50k": (768, 1020, 50000, "cholesky", 1.0),       # config 4
    "uhd200k": (2160, 3840, 200000, "cholesky", 1.0),      # config 5
    "kodak30k_fitted": (512, 768, 30000, "cholesky", 3.0),  # larger footprints (fitted-like)
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def make_inputs(name, device, seed=1):
    from gaussianimage_amd._testing import synth
    H, W, N, kind, fp = WORKLOADS[name]
    if kind == "cholesky":
        means, L, col, op = synth.cholesky_inputs(N, seed=seed, footprint=fp)
        args = (means, L, None, col, op.reshape(-1))
    else:
        means, sc, rot, col, op = synth.rs_inputs(N, seed=seed, footprint=fp)
        args = (means, sc, rot, col, op.reshape(-1))
    return tuple(None if a is None else a.to(device).contiguous() for a in args)


def algorithmic_bytes_raster(I, P, T):
    """Minimal traffic of one rasterize_sum forward launch: per intersection the Gaussian id
    (4 B) and its gathered xy (8), conic (12), opacity (4), colour (12); per pixel the 12 B
    written; per tile its two list bounds."""
    return 40 * I + 12 * P + 8 * T


def time_cpu_oracle(name, budget_s=12.0):
    """CPU baseline: this repository's C restatement (oracle) of the same forward path."""
    from gaussianimage_amd._testing import synth
    threads = min(16, os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    threads = int(os.environ["OMP_NUM_THREADS"])
    from oracle import cport
    H, W, N, kind, fp = WORKLOADS[name]
    if kind == "cholesky":
        means, L, col, op = synth.cholesky_inputs(N, seed=1, footprint=fp)
        a = (means.numpy(), L.numpy())
        proj = cport.project_cholesky_fwd
    else:
        means, sc, rot, col, op = synth.rs_inputs(N, seed=1, footprint=fp)
        a = (means.numpy(), sc.numpy(), rot.numpy())
        proj = cport.project_rs_fwd
    col, op = col.numpy(), op.numpy()

    def frame():
        xys, _, radii, conics, _ = proj(*a, H, W)
        offs, ids = cport.bin_tiles(xys, radii, H, W)
        return cport.rasterize_sum_fwd(xys, conics, col, op, H, W, offs, ids)

    def run(budget, most):
        frame()
        n, t0 = 0, time.perf_counter()
        while True:
            frame()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= most:
                return n, el

    n, el = run(budget_s, 1000)
    out = {"value": n / el, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{n} forward frames of {name} (project+bin+rasterize) in {el:.1f} s; "
                     "rasterize loop OpenMP-parallel over tiles, projection and binning serial"}
    try:  # the same on one core (a few frames), as a second reference point
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        n1, el1 = run(4.0, 20)
        gomp.omp_set_num_threads(threads)
        out["single_core_value"] = n1 / el1
    except OSError:
        pass
    return out


class no_gc:
    """Timed regions run with Python's cyclic garbage collector off (and a collection done up
    front): with torch imported a full collection takes several milliseconds -- longer than a
    whole timed region here -- and when one happens to fall into a region is a matter of the
    interpreter's allocation counters, not of the code being measured."""

    def __enter__(self):
        import gc
        self.was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was_enabled:
            gc.enable()


def bench_forward(pipe, inputs, steps, warmup, sync_all):
    """K forward frames, one C call each (gi2d_render_*_fwd).  The rasterize launch of every
    timed frame is bracketed by HIP events recorded inside that call on the launch stream."""
    means, cov_a, cov_b, col, op = inputs
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(steps)]
    for a, b in ev:  # give every event its handle outside the timed region
        a.record()
        b.record()
    run = pipe.bind(means, cov_a, cov_b, col, op)
    for _ in range(warmup):
        run()
    with no_gc():
        sync_all()
        t0 = time.perf_counter()
        for k in range(steps):
            # every 8th frame is bracketed: an event pair costs the GPU ~2 us of its own
            run(ev[k] if k % 8 == 0 else None)
        sync_all()
        el = time.perf_counter() - t0
    raster_ms = sorted(a.elapsed_time(b) for a, b in ev[::8])
    # What an event pair measures with NOTHING between the two records, on the same busy
    # stream (outside the timed region): the bracket's own share of a bracketed launch.
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(48)]
    for a, b in empty:
        run()
        a.record()
        b.record()
    sync_all()
    empty_ms = sorted(a.elapsed_time(b) for a, b in empty)
    return el, raster_ms, empty_ms[len(empty_ms) // 2]


def bench_region(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    with no_gc():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


def bench_model_loops(name, device, steps):
    """The reference's own two loops on a stand-in of its model (gaussianimage_amd/_testing/harness.py),
    through the drop-in operators: train_iter (forward, L2 loss, backward, PSNR .item(), Adan
    step, StepLR -- /root/reference/train.py:141-145) and the FPS loop (100 x model() under
    eval + no_grad, wall clock -- /root/reference/train.py:242-249; synchronised at both ends
    here)."""
    from gaussianimage_amd._testing import harness, synth
    H, W, N, kind, _ = WORKLOADS[name]
    torch.manual_seed(1)
    if kind == "cholesky":
        model = harness.CholeskyImageModel(N, H, W, device)
    else:
        model = harness.RSImageModel(N, H, W, device)
    gt = synth.target_image(H, W).to(device)
    out = {}
    model.train()
    for _ in range(10):
        model.train_iter(gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, psnr = model.train_iter(gt)
    torch.cuda.synchronize()
    out["train_step_ms"] = (time.perf_counter() - t0) / steps * 1e3
    out["train_step_note"] = ("reference-style train_iter: model forward (3 raster passes for "
                              "cholesky), mse, backward, .item(), fused Adan, StepLR")
    out["psnr_after_%d_iters" % (steps + 10)] = psnr
    model.eval()
    with torch.no_grad():
        for _ in range(10):
            model()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            model()
        torch.cuda.synchronize()
        out["reference_fps_loop"] = 100 / (time.perf_counter() - t0)
    if kind == "cholesky":
        # the same three images from one multi-set launch instead of three passes
        model.passes = "fused"
        with torch.no_grad():
            for _ in range(10):
                model()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                model()
            torch.cuda.synchronize()
            out["reference_fps_loop_fused_passes"] = 100 / (time.perf_counter() - t0)
        model.passes = 1
        model.train()
        for _ in range(5):
            model.train_iter(gt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.train_iter(gt)
        torch.cuda.synchronize()
        out["train_step_1pass_ms"] = (time.perf_counter() - t0) / steps * 1e3
    return out


def bench_decode(name, device, steps):
    """The codec's decoding-speed loop (/root/reference/test_quantize.py:69-74: 100 x
    decompress_wo_ec(encoding_dict), wall clock): the fused two-launch decode, and the same
    decode written the reference's way -- torch expressions for the de-quantisation and the
    code-vector lookup, the drop-in operators, clamp / view / permute."""
    from gaussianimage_amd import codec, ops
    H, W, N, kind, fp = WORKLOADS[name]
    enc, q = codec.synthetic_encoding(N, kind, seed=1, footprint=fp)
    dec = codec.CodecDecoder(enc, q, H, W, kind=kind, device=device)
    out = {}

    def loop(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    out["decode_fps_fused"] = loop(dec.decode, max(steps, 100))
    xyz, idx, cb = dec.xyz, dec.index.long(), dec.codebooks
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    bound = torch.tensor([0.5, 0, 0.5] if kind == "cholesky" else [0.5, 0.5],
                         device=device).view(1, -1)

    def reference_style():
        means = torch.tanh(xyz.float())
        colors = 0
        for i in range(cb.shape[0]):
            colors = colors + cb[i][idx[:, i]]
        if kind == "cholesky":
            L = dec.code_a * dec.scale_a + dec.beta_a + bound
            proj = ops.project_gaussians_2d(means, L, H, W, tb)
        else:
            sc = torch.abs(dec.code_a * dec.scale_a + dec.beta_a + bound)
            rot = dec.code_b * dec.scale_b + dec.beta_b
            proj = ops.project_gaussians_2d_scale_rot(means, sc, rot, H, W, tb)
        img = ops.rasterize_gaussians_sum(*proj, colors, dec.opacity, H, W, 16, 16,
                                          background=dec.background, return_alpha=False)
        img = torch.clamp(img, 0, 1)
        return img.view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()

    with torch.no_grad():
        out["decode_fps_reference_style_loop"] = loop(reference_style, 100)
    return out


def bench_concurrent(name, inputs, device, steps, streams=4):
    """Throughput with several independent frames in flight: `streams` pipelines, each with its
    own workspace and HIP stream (default priority), frames issued round-robin.  A single
    frame's kernels are latency-bound and leave most of the chip idle; independent frames
    (a batch of images, video frames) overlap.  The headline `value` stays the one-stream rate."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N, kind, _ = WORKLOADS[name]
    runs = []
    for _ in range(streams):
        st = torch.cuda.Stream(device=device)
        with torch.cuda.stream(st):
            run = FramePipeline(N, H, W, kind=kind, device=device).bind(*inputs)
            for _ in range(5):
                run()
        runs.append(run)
    # (the first few hundred overlapping launches of a process run well below the steady rate:
    # warm the streams up together, not only one by one)
    for k in range(400):
        runs[k % streams]()
    with no_gc():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            runs[k % streams]()
        torch.cuda.synchronize()
        out = {"fps_%d_streams" % streams: steps / (time.perf_counter() - t0)}
    # the same for ONE caller through the packaged ring (two frames in flight)
    from gaussianimage_amd.pipeline import FrameRing
    for depth in (2, 4):
        ring = FrameRing(N, H, W, kind=kind, device=device, depth=depth)
        for _ in range(200):
            ring.render(*inputs)
        with no_gc():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ring.render(*inputs, wait_inputs=False)
            torch.cuda.synchronize()
            out["fps_frame_ring_depth%d" % depth] = steps / (time.perf_counter() - t0)
        del ring
    return out


def bench_fused_fit(name, device, steps, concurrent=True):
    """The fused fitting iteration (gaussianimage_amd/fit.py): forward, L2 loss, backward and
    Adan step in two launches on the raw parameters, no host synchronisation inside the loop.
    The workload's footprint scales the initial Cholesky factors as it does the forward inputs."""
    from gaussianimage_amd._testing import synth
    from gaussianimage_amd.fit import FusedFitter
    H, W, N, kind, fp = WORKLOADS[name]

    def make(spatial_order=False):
        f = FusedFitter(N, H, W, kind=kind, device=device, keep_image=False,
                        spatial_order=spatial_order)
        if fp != 1.0 and kind == "cholesky":
            bound = torch.tensor([0.5, 0.0, 0.5], device=device)
            f.params["_cholesky"].copy_((f.params["_cholesky"] + bound) * fp - bound)
        return f

    torch.manual_seed(1)
    fit = make()
    gt = synth.target_image(H, W).to(device)
    for _ in range(10):
        fit.step(gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fit.step(gt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"fused_train_step_ms": ms, "fused_train_iters_per_s": 1e3 / ms,
           "fused_train_psnr_after_%d_iters" % (steps + 10): fit.psnr()}
    # the same with the Gaussians kept in tile order (shared list atomics)
    torch.manual_seed(1)
    fit = make(spatial_order=True)
    for _ in range(10):
        fit.step(gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fit.step(gt)
    torch.cuda.synchronize()
    out["fused_train_step_ms_spatial_order"] = (time.perf_counter() - t0) / steps * 1e3
    del fit
    if not concurrent:
        return out
    # several images fitted at once (the reference fits the images of a dataset one after the
    # other, /root/reference/train.py:306-316): one fitter per HIP stream, default priority
    streams = 4
    fits = []
    for _ in range(streams):
        st = torch.cuda.Stream(device=device)
        with torch.cuda.stream(st):
            fits.append(make())
    torch.cuda.synchronize()
    for _ in range(5):
        for f in fits:
            f.step(gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for f in fits:
            f.step(gt)
    torch.cuda.synchronize()
    out["fused_train_iters_per_s_%d_streams" % streams] = streams * steps / (time.perf_counter() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="kodak30k", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0,
                    help="seconds of host work spent on the CPU baseline sample")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true",
                    help="skip the multi-stream extras and the 4K point (overlapping kernels and "
                         "a second workload blur a profiler's per-kernel averages)")
    ap.add_argument("--raster-ppl", type=int, default=0, choices=[0, 1, 2, 4],
                    help="pixels per lane of the forward rasteriser (0 = the library's choice)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run "
                             "(--nproc-per-node N)")
    distributed = world > 1
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        # GI2D_BENCH_BACKEND=gloo rehearses the multi-rank path where RCCL cannot run (e.g. two
        # ranks on one GPU); collectives then go through host memory
        backend = os.environ.get("GI2D_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    from gaussianimage_amd.pipeline import FramePipeline

    # every measurement below runs with the cyclic garbage collector off (see no_gc); the
    # process is short-lived
    no_gc().__enter__()
    H, W, N, kind, fp = WORKLOADS[args.workload]
    inputs = make_inputs(args.workload, device, seed=1 + rank)
    pipe = FramePipeline(N, H, W, kind=kind, device=device)
    if args.raster_ppl:
        pipe.ws.set_raster_ppl(args.raster_ppl)
    pipe.forward(*inputs)
    # intersections as the reference counts them: the sum of num_tiles_hit (every tile of the
    # radius box), whatever part of them the lists actually hold after culling
    I = int(pipe.num_tiles_hit.sum().item())
    P, T = H * W, pipe.tiles_x * pipe.tiles_y

    el, raster_ms, empty_bracket_ms = bench_forward(pipe, inputs, args.steps, args.warmup,
                                                    sync_all)
    if distributed:
        t = torch.tensor([el], dtype=torch.float64,
                         device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = el / args.steps * 1e3
    fps = world * args.steps / el

    raster_avg_ms = sum(raster_ms) / len(raster_ms)
    bytes_raster = algorithmic_bytes_raster(I, P, T)
    achieved = bytes_raster / (raster_avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get("rasterize_sum_fwd_kernel")
        except Exception:
            traffic = None
    # wave-level VALU instructions of the rasteriser launch (SQ_INSTS_VALU from a counters-only
    # rocprofv3 pass of this workload, kept under profiles/) against the chip's issue rate:
    # 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
    valu = None
    mpath = os.path.join(ROOT, "profiles", "r01_sq_instruction_mix.json")
    if os.path.exists(mpath):
        try:
            mix = json.load(open(mpath)).get(args.workload, {})
            for name, c in mix.items():
                if "rasterize_sum_fwd_kernel<true, false, 1" in name:
                    peak = 256 * 4 * 2.4e9 / 4
                    valu = {"instr_per_launch": c["SQ_INSTS_VALU"], "peak_instr_per_s": peak,
                            "frac": c["SQ_INSTS_VALU"] / peak / (raster_avg_ms * 1e-3)}
        except Exception:
            valu = None
    result = {
        "metric": "render FPS (forward: project + tile binning + rasterize_sum), 768x512 @ 30k "
                  "Gaussians" if args.workload == "kodak30k" else f"render FPS, {args.workload}",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {W}x{H}, {N} Gaussians, {kind} covariance, "
                               f"model-init distribution seed {1 + rank}, footprint x{fp}",
                   "intersections": I, "tiles": T,
                   "parallelism": "1 process/GPU, independent frames per rank"},
        "roofline": {"bound": "hbm", "kernel": "rasterize_sum_fwd_kernel", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": bytes_raster,
                     "avg_launch_ms": raster_avg_ms,
                     "median_launch_ms": raster_ms[len(raster_ms) // 2],
                     "p10_launch_ms": raster_ms[len(raster_ms) // 10],
                     "p90_launch_ms": raster_ms[(9 * len(raster_ms)) // 10],
                     "empty_event_bracket_ms": empty_bracket_ms, "valu_issue": valu},
    }

    # (the secondary figures are single-GPU figures: the one-rank run carries them)
    if not args.no_extras and rank == 0 and world == 1:
        ex = {}
        means, cov_a, cov_b, col, op = inputs
        v_img = torch.randn(H, W, 3, device=device)

        def fwd_bwd():
            pipe.forward(means, cov_a, cov_b, col, op)
            pipe.backward(means, cov_a, cov_b, col, op, v_img)

        k = max(20, args.steps // 2)
        # the headline loop without its event brackets (each bracketed frame costs ~6 us more)
        run_plain = pipe.bind(means, cov_a, cov_b, col, op)
        ex["forward_fps_no_event_brackets"] = 1e3 / bench_region(run_plain, max(args.steps, 100), 10)
        ex["fwd_bwd_ms"] = bench_region(fwd_bwd, k, 5)
        ex["project_ms"] = bench_region(lambda: pipe.project(means, cov_a, cov_b), k, 5)
        pipe.project(means, cov_a, cov_b)
        pipe.bin()

        def proj_bin():
            pipe.project(means, cov_a, cov_b)
            pipe.bin()

        ex["project_bin_ms"] = bench_region(proj_bin, k, 5)
        ex["rasterize_ms"] = bench_region(lambda: pipe.rasterize(col, op), k, 5)
        three = [col, torch.rand_like(col) * 0.5, torch.ones_like(col)]
        ex["rasterize_3sets_one_launch_ms"] = bench_region(
            lambda: pipe.rasterize_multi(three, op), k, 5)
        ex["rasterize_3sets_three_launches_ms"] = bench_region(
            lambda: [pipe.rasterize(c, op) for c in three], k, 5)
        ex["bwd_only_ms"] = bench_region(
            lambda: pipe.backward(means, cov_a, cov_b, col, op, v_img), k, 5)
        # The same Gaussians stored in the order of the tiles their centres fall in (what a
        # fitter or an encoder that keeps a spatial order hands over): the lanes of a wave then
        # share their list atomics.  The headline keeps the model's own random order.
        with torch.no_grad():
            tcol = ((0.5 * W * means[:, 0] + 0.5 * W) / 16).floor().clamp_(0, (W - 1) // 16)
            trow = ((0.5 * H * means[:, 1] + 0.5 * H) / 16).floor().clamp_(0, (H - 1) // 16)
            order = torch.argsort(trow * ((W + 15) // 16) + tcol, stable=True)
            sorted_in = tuple(None if t is None else t[order].contiguous() for t in inputs)
        pipe_s = FramePipeline(N, H, W, kind=kind, device=device)
        run_s = pipe_s.bind(*sorted_in)
        ex["forward_fps_spatial_order"] = 1e3 / bench_region(run_s, max(k, 100), 10)
        del pipe_s, run_s
        # Pair evaluations per second against the fp32 vector peak: the inner loop is VALU
        # and transcendental work on cache-resident data, not HBM traffic.
        pairs = I * 256
        ex["pair_evals_per_s"] = pairs / (raster_avg_ms * 1e-3)
        if not args.no_concurrent:
            ex.update(bench_concurrent(args.workload, inputs, device, max(args.steps, 200)))
        ex.update(bench_model_loops(args.workload, device, k))
        if args.workload == "kodak30k" and not args.no_concurrent:
            # the one-GPU point of the tile-row scaling series (3840x2160 / 200k Gaussians)
            big = FramePipeline(*[WORKLOADS["uhd200k"][i] for i in (2, 0, 1)],
                                kind=WORKLOADS["uhd200k"][3], device=device)
            big_in = make_inputs("uhd200k", device, seed=1)
            big_v = torch.randn(WORKLOADS["uhd200k"][0], WORKLOADS["uhd200k"][1], 3, device=device)
            ex["uhd200k_1gpu_forward_ms"] = bench_region(lambda: big.forward(*big_in), 50, 5)
            ex["uhd200k_1gpu_fwd_bwd_ms"] = bench_region(
                lambda: (big.forward(*big_in), big.backward(*big_in, big_v)), 50, 5)
            del big, big_in, big_v
        ex.update(bench_decode(args.workload, device, k))
        ex.update(bench_fused_fit(args.workload, device, max(k, 100),
                                  concurrent=not args.no_concurrent))
        result["extras"] = ex
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        result["cpu_baseline"] = time_cpu_oracle(args.workload, budget_s=args.cpu_budget_s)
        result["cpu_baseline"]["gpu_over_cpu"] = fps / result["cpu_baseline"]["value"]

    if distributed:
        # Secondary measurement: one 4K frame split by tile rows.  It runs in a watched thread:
        # a collective that never completes (or a rank that died) must not cost the headline
        # line, so after GI2D_SHARD_TIMEOUT_S the ranks stop waiting, rank 0 prints what it has,
        # and every rank leaves without joining the stuck work.
        import threading
        box = {}

        def sharded():
            try:
                torch.cuda.set_device(local_rank)
                from gaussianimage_amd.sharding import bench_tile_row_sharded
                box["result"] = bench_tile_row_sharded(device, rank, world,
                                                       steps=max(20, args.steps // 4), warmup=5)
                dist.barrier()
                box["done"] = True
            except Exception as e:
                box["result"] = {"error": repr(e)}

        th = threading.Thread(target=sharded, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("GI2D_SHARD_TIMEOUT_S", "150")))
        if th.is_alive():
            box.setdefault("result", {"error": "timed out waiting for the ranks"})
        if rank == 0:
            result["tile_row_sharded"] = box.get("result", {"error": "no result"})
            print(json.dumps(result), flush=True)
        sys.stdout.flush()
        if box.get("done"):
            dist.destroy_process_group()
            return
        os._exit(0)  # stuck or failed collective: do not wait for it at interpreter exit
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
