#!/usr/bin/env python3
"""Benchmark of the hot path: project -> tile binning -> rasterize_sum (forward), through the
C ABI, on synthetic parameters with the reference models' own initial distributions.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--batch B]

One "step" is one pass of the hot path over one BATCH of B independent frames (default 24 -- the
size of the Kodak set the reference loops over, /root/reference/train.py:358-366 -- each with
its own Gaussians) whose operator inputs are already resident in HBM: one C call
(gi2d_render_batch), two launches.  N=1 runs BASELINE config 2 (768x512, 30k Gaussians,
Cholesky).  `value` is frames per second.  The timed region is EXACTLY K steps between
synchronisations; because K steps of this path last only about a millisecond, that region is
repeated until at least `--min-time-ms` have been timed and the median region is reported
(`timed_regions`, `frames_timed` say how much that was).
For N>1 (launched under torch.distributed.run, one rank per GPU) every rank renders its own
batches -- frames are independent units, so there is no data-path collective and scaling is weak
-- and, as a secondary measurement reported under ``tile_row_sharded``, the ranks render ONE
3840x2160 / 200k frame split by tile rows.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      -- the dominant kernel (rasterize_sum forward) against the HBM roofline, its
                   launch time taken from HIP events the library records around the launch on
                   the launch stream, inside the timed regions,
  kernels       -- the same for every kernel of the forward, backward and fused fitting step,
  train_step    -- the other half of the metric: ms per training iteration (reference-style
                   loop, fused step, forward+backward alone),
  cpu_baseline  -- this repository's CPU oracle timed on the host cores (a reported
                   baseline; the reference has no CPU path),
  extras        -- single-frame latency-bound rates, stages, decode, concurrency.
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
# wave64 vector instructions per second: 256 CUs x 4 SIMDs x 2.4 GHz, 2 cycles each once a SIMD
# holds two or more waves (tools/micro/valu_issue.hip measures 1.04-1.07e12 for streams of
# v_fma_f32 at the 2.2-2.35 GHz the part holds; one wave alone issues every 4.9 cycles)
VALU_PEAK_INSTR_S = 256 * 4 * 2.4e9 / 2


def tiles_of(H, W):
    return ((W + 15) // 16) * ((H + 15) // 16)


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


def make_batch_inputs(name, device, frames, seed=1):
    """`frames` different frames of the workload as contiguous [frames, N, ...] tensors."""
    per = [make_inputs(name, device, seed=seed + f) for f in range(frames)]
    return tuple(None if per[0][k] is None else torch.stack([p[k] for p in per]).contiguous()
                 for k in range(5))


def bound_batch(name, device, frames, seed=1):
    """(FrameBatch, run()) for `frames` different frames of a workload."""
    from gaussianimage_amd.pipeline import FrameBatch
    H, W, N, kind, _ = WORKLOADS[name]
    fb = FrameBatch(frames, N, H, W, kind=kind, device=device)
    return fb, fb.bind(*make_batch_inputs(name, device, frames, seed=seed))


# ---- algorithmic bytes (minimal traffic of an ideal implementation, fp32, 3 channels) ----------
def algorithmic_bytes_raster(I, P, T):
    """rasterize_sum forward: per intersection the Gaussian id (4 B) and its gathered xy (8),
    conic (12), opacity (4), colour (12); per pixel the 12 B written; per tile its list bounds."""
    return 40 * I + 12 * P + 8 * T


def algorithmic_bytes(kernel, N, I, P, T):
    """Per launch, for N Gaussians, I (tile, Gaussian) intersections, P pixels, T tiles."""
    return {
        # read means + covariance parameters, write the five outputs, one id + one counter word
        # per intersection
        "project": 20 * N + 32 * N + 8 * I,
        "raster_fwd": algorithmic_bytes_raster(I, P, T),
        # read v_out, re-gather the attributes, write nine partial sums + their stamp
        "raster_bwd": 12 * P + 40 * I + 40 * I,
        # read the partial sums, xys and radii; write v_xy, v_conic, v_colors, v_opacity
        "gather": 40 * I + 12 * N + 36 * N,
        # forward + loss + backward pass 1 in one kernel: target image in, partial sums out
        "fit_raster": 12 * P + 40 * I + 40 * I + 4 * T,
        # partial sums in; 8 parameters x (p, m, n, d, prev) read and written; the next frame's
        # projection outputs and list entries out
        "fit_tail": 40 * I + 8 * 4 * 5 * 2 * N + 32 * N + 8 * I,
    }[kernel]


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
           "host_cores": os.cpu_count(), "host_cpu": host_cpu_model(),
           "sample": f"{n} forward frames of {name} (project+bin+rasterize) in {el:.1f} s on "
                     f"{threads} OpenMP threads (the box's share of a {os.cpu_count()}-CPU host): "
                     "projection parallel over Gaussians, binning over bands of tile rows, "
                     "rasterisation over tiles"}
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


class Probe:
    """HIP-event brackets around one kind of kernel launch, recorded by the library itself on the
    launch stream (gi2d_workspace_set_probe): `arm()` before a call that contains the launch,
    `disarm()` after it; `ms()` reads the durations once everything has completed."""

    def __init__(self, ws, kernel, pairs):
        """`ws`: a workspace, or a callable that returns the workspace the NEXT call will use."""
        from gaussianimage_amd import _lib
        self._lib, self.ws, self.kernel = _lib, ws, _lib.PROBE[kernel]
        self.armed = None
        self.pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                      for _ in range(pairs)]
        for a, b in self.pairs:   # a torch event gets its handle when first recorded
            a.record()
            b.record()
        self.used = 0

    def arm(self):
        a, b = self.pairs[self.used]
        self.used += 1
        self.armed = self.ws() if callable(self.ws) else self.ws
        self._lib.lib.gi2d_workspace_set_probe(self.armed.handle, self.kernel, a.cuda_event,
                                               b.cuda_event)

    def disarm(self):
        if self.armed is not None:
            self._lib.lib.gi2d_workspace_set_probe(self.armed.handle, 0, None, None)
            self.armed = None

    def ms(self):
        return sorted(a.elapsed_time(b) for a, b in self.pairs[:self.used])


def probe_kernel(ws, kernel, fn, reps=48, warmup=4):
    """Average launch duration (ms) of `kernel` over `reps` calls of fn() -- outside the headline's
    timed regions; every call is bracketed."""
    for _ in range(warmup):
        fn()
    p = Probe(ws, kernel, reps)
    for _ in range(reps):
        p.arm()
        fn()
    p.disarm()
    torch.cuda.synchronize()
    ms = p.ms()
    return {"avg_launch_ms": sum(ms) / len(ms), "median_launch_ms": ms[len(ms) // 2],
            "launches": len(ms)}


def roofline_entry(kernel, timing, N, I_8d, I_held, P, T):
    """HBM-roofline entry of one kernel: algorithmic bytes with the intersections as the
    reference counts them (sum of num_tiles_hit), and with the list entries actually held."""
    b8d = algorithmic_bytes(kernel, N, I_8d, P, T)
    bheld = algorithmic_bytes(kernel, N, I_held, P, T)
    t = timing["avg_launch_ms"] * 1e-3
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": b8d, "achieved": b8d / t / 1e9,
            "frac": b8d / t / 1e9 / HBM_PEAK_GBS,
            "bytes_with_lists_held": bheld, "achieved_lists_held": bheld / t / 1e9,
            "frac_lists_held": bheld / t / 1e9 / HBM_PEAK_GBS, **timing}


def bench_regions(run, steps, warmup, sync_all, min_time_s, probe=None, probe_every=4,
                  all_agree=lambda done: done):
    """Times regions of EXACTLY `steps` calls of run(), each between synchronisations, until
    `min_time_s` seconds have been timed (at least 3 regions, at most 400).  Returns the list of
    region times.  `probe`: bracket every `probe_every`-th step's dominant launch.
    `all_agree(done)`: with several ranks, true only when every rank is done (the ranks must
    leave the loop together: its synchronisations are collective)."""
    for _ in range(warmup):
        run()
    regions = []
    with no_gc():
        while True:
            sync_all()
            t0 = time.perf_counter()
            for k in range(steps):
                if probe is not None and k % probe_every == 0 and probe.used < len(probe.pairs):
                    probe.arm()
                    run()
                    probe.disarm()
                else:
                    run()
            # (the clock stops when this rank's own work has completed; the barrier that closes
            # the bracket follows, and the slowest rank's time is what gets reported)
            torch.cuda.synchronize()
            regions.append(time.perf_counter() - t0)
            sync_all()
            if all_agree(len(regions) >= 400 or
                         (len(regions) >= 3 and sum(regions) >= min_time_s)):
                return regions


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
    """The reference's own two loops on a stand-in of its model
    (gaussianimage_amd/_testing/harness.py), through the drop-in operators: train_iter (forward,
    L2 loss, backward, PSNR .item(), Adan step, StepLR -- /root/reference/train.py:141-145) and
    the FPS loop (100 x model() under eval + no_grad, wall clock --
    /root/reference/train.py:242-249; synchronised at both ends here)."""
    from gaussianimage_amd import ops
    from gaussianimage_amd._testing import harness, synth
    H, W, N, kind, _ = WORKLOADS[name]
    torch.manual_seed(1)
    if kind == "cholesky":
        model = harness.CholeskyImageModel(N, H, W, device)
    else:
        model = harness.RSImageModel(N, H, W, device)
    gt = synth.target_image(H, W).to(device)
    out = {"operator_binding": "compiled (csrc/torch_ops.cpp)" if ops.compiled_binding()
           else "ctypes (ops.py)"}
    if ops.compiled_binding():
        # the same two loops through the ctypes binding of the same kernels, for comparison
        prev = ops.use_compiled(False)
        try:
            for _ in range(10):
                model.train_iter(gt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model.train_iter(gt)
            torch.cuda.synchronize()
            out["train_step_ms_ctypes_binding"] = (time.perf_counter() - t0) / steps * 1e3
            with torch.no_grad():
                for _ in range(10):
                    model()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    model()
                torch.cuda.synchronize()
                out["reference_fps_loop_ctypes_binding"] = 100 / (time.perf_counter() - t0)
        finally:
            ops.use_compiled(prev)
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
    own workspace and HIP stream (default priority), frames issued round-robin; and the packaged
    ring for one caller.  (The batched call of the headline gets the same overlap by design, in
    one launch pair, without depending on which hardware queues the streams land on.)"""
    from gaussianimage_amd.pipeline import FramePipeline, FrameRing
    H, W, N, kind, _ = WORKLOADS[name]
    runs = []
    for _ in range(streams):
        st = torch.cuda.Stream(device=device)
        with torch.cuda.stream(st):
            run = FramePipeline(N, H, W, kind=kind, device=device).bind(*inputs)
            for _ in range(5):
                run()
        runs.append(run)
    for k in range(400):
        runs[k % streams]()
    with no_gc():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            runs[k % streams]()
        torch.cuda.synchronize()
        out = {"fps_%d_streams" % streams: steps / (time.perf_counter() - t0)}
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


def make_fitter(name, device, spatial_order=False):
    from gaussianimage_amd.fit import FusedFitter
    H, W, N, kind, fp = WORKLOADS[name]
    f = FusedFitter(N, H, W, kind=kind, device=device, keep_image=False,
                    spatial_order=spatial_order)
    if fp != 1.0 and kind == "cholesky":
        bound = torch.tensor([0.5, 0.0, 0.5], device=device)
        f.params["_cholesky"].copy_((f.params["_cholesky"] + bound) * fp - bound)
    return f


def bench_fused_fit(name, device, steps, concurrent=True):
    """The fused fitting iteration (gaussianimage_amd/fit.py): forward, L2 loss, backward and
    Adan step in two launches on the raw parameters, no host synchronisation inside the loop.
    The workload's footprint scales the initial Cholesky factors as it does the forward inputs.
    Returns (figures, fitter, target)."""
    from gaussianimage_amd._testing import synth
    H, W, N, kind, fp = WORKLOADS[name]
    torch.manual_seed(1)
    fit = make_fitter(name, device)
    gt = synth.target_image(H, W).to(device)
    ms = bench_region(lambda: fit.step(gt), steps, 10)
    out = {"fused_train_step_ms": ms, "fused_train_iters_per_s": 1e3 / ms,
           "fused_train_psnr_after_%d_iters" % (steps + 10): fit.psnr()}
    torch.manual_seed(1)
    fit_s = make_fitter(name, device, spatial_order=True)
    out["fused_train_step_ms_spatial_order"] = bench_region(lambda: fit_s.step(gt), steps, 10)
    del fit_s
    if concurrent:
        # several images fitted at once (the reference fits the images of a dataset one after
        # the other, /root/reference/train.py:306-316): one fitter per HIP stream
        streams = 4
        fits = []
        for _ in range(streams):
            st = torch.cuda.Stream(device=device)
            with torch.cuda.stream(st):
                fits.append(make_fitter(name, device))
        torch.cuda.synchronize()
        ms4 = bench_region(lambda: [f.step(gt) for f in fits], steps, 5)
        out["fused_train_iters_per_s_%d_streams" % streams] = streams * 1e3 / ms4
    return out, fit, gt


def bench_full_fit(device, iters=50000):
    """BASELINE config 4 as a whole: the reference's default fit -- 50 000 iterations
    (/root/reference/train.py:141-145,287), 50 000 Gaussians, lr 1e-3 with StepLR(20000, 0.5) --
    of a ~1020x768 synthetic target through the fused fitting step; the host synchronises only at
    the PSNR read-outs."""
    from gaussianimage_amd._testing import synth
    H, W, N, kind, _ = WORKLOADS["div2k50k"]
    torch.manual_seed(1)
    fit = make_fitter("div2k50k", device)
    gt = synth.target_image(H, W).to(device)
    marks = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, iters + 1):
        fit.step(gt)
        if it in (1000, 10000, iters):
            marks[str(it)] = round(fit.psnr(), 3)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"workload": f"div2k50k: {W}x{H}, {N} Gaussians, {kind}", "iterations": iters,
            "wall_s": wall, "ms_per_iteration": wall / iters * 1e3, "psnr_db_at": marks,
            "overflowed_frames": fit.overflowed_frames,
            "list_capacity": fit.ws.info()["capacity"]}


def secondary(args, device, batch, run, batch_in, B, I_8d, I_held, raster_batch_ms):
    """Per-kernel roofline entries, the training-step half of the metric, and the extras."""
    from gaussianimage_amd.pipeline import FramePipeline
    H, W, N, kind, fp = WORKLOADS[args.workload]
    P, T = H * W, tiles_of(H, W)
    k = max(20, args.steps // 2)
    out, ex, kernels = {}, {}, {}

    # ---- single frame (the latency-bound rate: two launches per frame, one frame in flight) ----
    inputs = tuple(None if t is None else t[0].contiguous() for t in batch_in)
    means, cov_a, cov_b, col, op = inputs
    pipe = FramePipeline(N, H, W, kind=kind, device=device)
    if args.raster_ppl:
        pipe.ws.set_raster_ppl(args.raster_ppl)
    run1 = pipe.bind(*inputs)
    ex["single_frame_fps"] = 1e3 / bench_region(run1, max(args.steps, 200), 20)
    I1_8d = int(pipe.num_tiles_hit.sum().item())
    I1_held = pipe.intersections()
    run1()
    v_img = torch.randn(H, W, 3, device=device)

    def bwd():
        pipe.backward(means, cov_a, cov_b, col, op, v_img)

    def fwd_bwd():
        run1()
        bwd()

    # ---- per-kernel launch times (library-recorded HIP events) and roofline fractions ----------
    kernels["project (batch of %d)" % B] = roofline_entry(
        "project", probe_kernel(batch.ws, "project", run), B * N, I_8d, I_held, B * P, B * T)
    kernels["project (1 frame)"] = roofline_entry(
        "project", probe_kernel(pipe.ws, "project", run1), N, I1_8d, I1_held, P, T)
    kernels["raster_fwd (1 frame)"] = roofline_entry(
        "raster_fwd", probe_kernel(pipe.ws, "raster_fwd", run1), N, I1_8d, I1_held, P, T)
    kernels["raster_bwd (1 frame)"] = roofline_entry(
        "raster_bwd", probe_kernel(pipe.ws, "raster_bwd", fwd_bwd), N, I1_8d, I1_held, P, T)
    kernels["gather (1 frame)"] = roofline_entry(
        "gather", probe_kernel(pipe.ws, "gather", fwd_bwd), N, I1_8d, I1_held, P, T)

    # ---- the training step: the other half of the metric ----------------------------------------
    fwd_bwd_ms = bench_region(fwd_bwd, k, 5)
    fit_figs, fit, gt = bench_fused_fit(args.workload, device, max(k, 100),
                                        concurrent=not args.no_concurrent)
    fit.forward(gt)   # (a plain forward: the lists of the current parameters, finalised)
    I_fit = int(fit.ws.offsets()[-1].item())
    I_fit_8d = int(fit.num_tiles_hit.sum().item())
    kernels["fit_raster (1 frame)"] = roofline_entry(
        "fit_raster", probe_kernel(fit.ws, "fit_raster", lambda: fit.step(gt)), N, I_fit_8d,
        I_fit, P, T)
    kernels["fit_tail (1 frame)"] = roofline_entry(
        "fit_tail", probe_kernel(fit.ws, "fit_tail", lambda: fit.step(gt)), N, I_fit_8d, I_fit,
        P, T)
    model_figs = bench_model_loops(args.workload, device, k)
    # bytes per iteration: forward (projection + lists + gathers + image), backward (v_out,
    # re-gather, partial sums, per-Gaussian gradients, projection backward), and the Adan
    # step's single-pass traffic over the 8 N parameter floats
    b_fwd = 20 * N + 32 * N + 4 * I1_8d + 8 * T + 40 * I1_8d + 12 * P
    b_bwd = 12 * P + 40 * I1_8d + 36 * I1_8d + 36 * N + 40 * N
    b_adan = 8 * N * 4 * 11
    fused_ms = fit_figs["fused_train_step_ms"]
    out["train_step"] = {
        "unit": "ms per iteration (forward + L2 loss + backward + Adan), 1 frame",
        "fused_ms": fused_ms, "fused_iters_per_s": 1e3 / fused_ms,
        "reference_style_ms": model_figs["train_step_ms"],
        "reference_style_note": model_figs["train_step_note"],
        "fwd_bwd_ms": fwd_bwd_ms,
        "algorithmic_bytes": {"forward": b_fwd, "backward": b_bwd, "adan": b_adan},
        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "achieved": (b_fwd + b_bwd + b_adan) / (fused_ms * 1e-3) / 1e9,
                     "frac": (b_fwd + b_bwd + b_adan) / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "what": "the whole fused step (2 launches): forward + backward + Adan "
                             "bytes / step time"},
        "fwd_bwd_roofline_frac": (b_fwd + b_bwd) / (fwd_bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    if T < 20000:
        # the same fused iteration for a batch of images fitted in lockstep (one call, three
        # launches: batch projection, fused rasterisation, tail): the images of a dataset
        from gaussianimage_amd._testing import synth
        from gaussianimage_amd.fit import BatchFitter
        gts = torch.stack([synth.target_image(H, W, seed=f)[0] for f in range(B)]).to(
            device).contiguous()
        bfit = BatchFitter(B, N, H, W, kind=kind, device=device)
        if fp != 1.0 and kind == "cholesky":   # (the workload's footprint, as in make_fitter)
            bound = torch.tensor([0.5, 0.0, 0.5], device=device)
            bfit.params["_cholesky"].copy_((bfit.params["_cholesky"] + bound) * fp - bound)
        bms = bench_region(lambda: bfit.step(gts), max(20, k // 2), 10)
        out["train_step"]["fused_batch"] = {
            "images": B, "ms_per_step": bms, "ms_per_image_iteration": bms / B,
            "image_iterations_per_s": B * 1e3 / bms,
            "roofline_frac": (b_fwd + b_bwd + b_adan) * B / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del bfit, gts
    out["kernels"] = kernels

    # ---- extras -------------------------------------------------------------------------------
    ex.update({k_: v for k_, v in fit_figs.items() if k_ != "fused_train_step_ms"})
    ex.update({k_: v for k_, v in model_figs.items() if k_ not in ("train_step_ms",
                                                                   "train_step_note")})
    ex["bwd_only_ms"] = bench_region(bwd, k, 5)
    three = [col, torch.rand_like(col) * 0.5, torch.ones_like(col)]
    pipe.project(means, cov_a, cov_b)
    pipe.bin()
    ex["rasterize_ms"] = bench_region(lambda: pipe.rasterize(col, op), k, 5)
    ex["rasterize_3sets_one_launch_ms"] = bench_region(lambda: pipe.rasterize_multi(three, op), k, 5)
    ex["rasterize_3sets_three_launches_ms"] = bench_region(
        lambda: [pipe.rasterize(c, op) for c in three], k, 5)
    # Pair evaluations per second (256 lanes per listed intersection): the inner loop is VALU
    # and transcendental work on cache-resident data, not HBM traffic.
    ex["pair_evals_per_s_batched"] = I_held * 256 / (raster_batch_ms * 1e-3)
    if not args.no_concurrent:
        ex.update(bench_concurrent(args.workload, inputs, device, max(args.steps, 200)))
        # the headline's batches with 2 and 3 of them in flight on streams of their own (the
        # projection of one under the rasteriser of another); the headline itself is one batch
        # at a time on one stream
        from gaussianimage_amd.pipeline import BatchRing
        for depth in (2, 3):
            ring = BatchRing(B, N, H, W, kind=kind, device=device, depth=depth)
            ms = bench_region(lambda: ring.render(*batch_in), 200, 40)
            ex["fps_batches_in_flight_%d" % depth] = B * 1e3 / ms
            del ring
        sweep = {}
        for b in (1, 2, 4, 8, 16, 32):   # the headline call at other batch sizes
            fb, frun = bound_batch(args.workload, device, b)
            sweep[str(b)] = b * 1e3 / bench_region(frun, 60, 10)
            del fb, frun
        ex["fps_by_batch_size"] = sweep
    if args.workload == "kodak30k" and not args.no_concurrent:
        # the one-GPU point of the tile-row scaling series (3840x2160 / 200k Gaussians)
        uh, uw, un, ukind, _ = WORKLOADS["uhd200k"]
        big = FramePipeline(un, uh, uw, kind=ukind, device=device)
        big_in = make_inputs("uhd200k", device, seed=1)
        big_v = torch.randn(uh, uw, 3, device=device)
        brun = big.bind(*big_in)
        ex["uhd200k_1gpu_forward_ms"] = bench_region(brun, 50, 5)
        ex["uhd200k_1gpu_fwd_bwd_ms"] = bench_region(
            lambda: (brun(), big.backward(*big_in, big_v)), 50, 5)
        Ib = int(big.num_tiles_hit.sum().item())
        Ibh = big.intersections()
        brun()
        kernels["raster_fwd (uhd200k, 1 frame)"] = roofline_entry(
            "raster_fwd", probe_kernel(big.ws, "raster_fwd", brun), un, Ib, Ibh, uh * uw,
            tiles_of(uh, uw))
        kernels["project (uhd200k, 1 frame)"] = roofline_entry(
            "project", probe_kernel(big.ws, "project", brun), un, Ib, Ibh, uh * uw,
            tiles_of(uh, uw))
        del big, big_in, big_v
    if args.workload == "kodak30k" and not args.no_concurrent:
        out["train_step"]["config4_full_fit"] = bench_full_fit(device)
    ex.update(bench_decode(args.workload, device, k))
    out["extras"] = ex
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n, argv, port):
    """The command line that runs this script as `n` ranks of one node, one rank per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
            str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__), *argv]


def launch_env(n, visible_gpus, env):
    """Environment of the ranks.  With fewer GPUs than ranks (a one-GPU development box) the ranks
    share devices, which RCCL cannot do: the collectives then go through gloo and host memory --
    a rehearsal of the multi-rank path whose line says so -- unless the caller chose a backend."""
    env = dict(env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if visible_gpus < n:
        env.setdefault("GI2D_BENCH_BACKEND", "gloo")
    return env


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: run the ranks as a child
    `python -m torch.distributed.run ... bench.py <same arguments>`, pass their output through
    and return their exit status.  This process makes no GPU call before or after (counting the
    devices does not initialise HIP)."""
    import subprocess
    cmd = launch_command(n, argv, free_port())
    env = launch_env(n, torch.cuda.device_count(), os.environ)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="kodak30k", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step (0 = 8, or 1 for frames of 20 000 tiles and more)")
    ap.add_argument("--min-time-ms", type=float, default=60.0,
                    help="keep timing K-step regions until this much has been timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0,
                    help="seconds of host work spent on the CPU baseline sample")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-concurrent", action="store_true",
                    help="skip the multi-stream extras and the 4K point (overlapping kernels and "
                         "a second workload blur a profiler's per-kernel averages)")
    ap.add_argument("--raster-ppl", type=int, default=0, choices=[0, 1, 2, 4],
                    help="pixels per lane of the forward rasteriser (0 = the library's choice)")
    ap.add_argument("--no-records", action="store_true",
                    help="the rasteriser gathers the operator tensors instead of per-Gaussian records")
    ap.add_argument("--project-form", type=int, default=0,
                    help="how the projection fills the lists (GI2D_OPT_PROJECT_AGG; 0 = the "
                         "library's choice)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="steps in flight: consecutive steps alternate between this many batch "
                         "objects, each on a stream of its own")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as `python bench.py --gpus N`: start the N ranks here (nothing in this process
        # has touched the GPU yet) and leave with their exit status
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}")
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

    from gaussianimage_amd.pipeline import FrameBatch

    # every measurement below runs with the cyclic garbage collector off (see no_gc); the
    # process is short-lived
    no_gc().__enter__()
    H, W, N, kind, fp = WORKLOADS[args.workload]
    P, T = H * W, tiles_of(H, W)
    # frames per step: the 24 images of the Kodak set (/root/reference/train.py:358-359) for the
    # small frames; a 3840x2160 frame fills the chip alone
    B = args.batch or (24 if T < 20000 else 1)
    batch_in = make_batch_inputs(args.workload, device, B, seed=1 + rank * B)
    batch = FrameBatch(B, N, H, W, kind=kind, device=device)
    if args.raster_ppl:
        batch.ws.set_raster_ppl(args.raster_ppl)
    if args.no_records:
        batch.ws.set_records(False)
    if args.project_form:
        from gaussianimage_amd import _lib
        _lib.check(_lib.lib.gi2d_workspace_set_option(batch.ws.handle, _lib.OPT_PROJECT_AGG,
                                                      args.project_form), "set_option")
    run = batch.bind(*batch_in)
    run()
    run()
    # intersections as the reference counts them: the sum of num_tiles_hit (every tile of the
    # radius box) -- and the (tile, Gaussian) pairs the lists actually hold after culling
    I_8d = int(batch.num_tiles_hit.sum().item())
    I_held = batch.intersections()
    run()

    def all_agree(done):
        if not distributed:
            return done
        flag = torch.tensor([1 if done else 0], dtype=torch.int32,
                            device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    # --in-flight D > 1: consecutive steps go to D batch objects, each with its own workspace,
    # outputs and (default-priority) stream, so that the projection of one step's batch runs
    # under the rasteriser of the step before; every step is still one call for one batch, and
    # a timed region still ends when all of its steps have completed.
    D = max(1, args.in_flight)
    step_run = run
    if D > 1:
        from gaussianimage_amd.pipeline import BatchRing
        ring = BatchRing(B, N, H, W, kind=kind, device=device, depth=D)
        if args.raster_ppl:
            for fb in ring.slots:
                fb.ws.set_raster_ppl(args.raster_ppl)

        def step_run():
            ring.render(*batch_in)

        for _ in range(3 * D):
            step_run()
        torch.cuda.synchronize()
        probe = Probe(lambda: ring.next_slot().ws, "raster_fwd", 256)
    else:
        probe = Probe(batch.ws, "raster_fwd", 256)
    regions = bench_regions(step_run, args.steps, args.warmup, sync_all, args.min_time_ms * 1e-3,
                            probe=probe, all_agree=all_agree)
    el = sorted(regions)[len(regions) // 2]
    if distributed:
        t = torch.tensor([el], dtype=torch.float64,
                         device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_per_step = el / args.steps * 1e3
    fps = world * B * args.steps / el

    raster_ms = probe.ms()
    timing = {"avg_launch_ms": sum(raster_ms) / len(raster_ms),
              "median_launch_ms": raster_ms[len(raster_ms) // 2],
              "p10_launch_ms": raster_ms[len(raster_ms) // 10],
              "p90_launch_ms": raster_ms[(9 * len(raster_ms)) // 10], "launches": len(raster_ms)}
    # (I_8d and I_held are totals over the batch: the launch covers B*N Gaussians, B*T tiles)
    roofline = roofline_entry("raster_fwd", timing, B * N, I_8d, I_held, B * P, B * T)
    roofline["kernel"] = "rasterize_sum_fwd_kernel (one launch renders the step's %d frames)" % B
    # HBM traffic and instruction counts of the same launch come from counters-only rocprofv3
    # passes (a separate run of this command; tools/gpu_profiles_r03.sh): kept with their source
    roofline["traffic"] = None
    cpath = os.path.join(ROOT, "profiles", "r03_counters.json")
    if os.path.exists(cpath):
        try:
            c = json.load(open(cpath)).get("%s_batch%d" % (args.workload, B), {})
            roofline["traffic"] = c.get("raster_fwd_hbm_bytes")
            roofline["traffic_source"] = c.get("source")
            if c.get("raster_fwd_valu_instr"):
                roofline["valu_issue"] = {
                    "instr_per_launch": c["raster_fwd_valu_instr"],
                    "peak_instr_per_s": VALU_PEAK_INSTR_S, "source": c.get("source"),
                    "frac": c["raster_fwd_valu_instr"] / VALU_PEAK_INSTR_S /
                    (timing["avg_launch_ms"] * 1e-3),
                    "note": "the peak is that of 2-cycle instructions (fma/mul/add); compares, "
                            "selects and SGPR-operand forms take 4 cycles, v_exp_f32 8 or more: "
                            "one pair evaluation is about 40 cycles "
                            "(profiles/r03_experiments/README.md)"}
        except Exception:
            pass

    result = {
        "metric": "render FPS (forward: project + tile binning + rasterize_sum), 768x512 @ 30k "
                  "Gaussians" if args.workload == "kodak30k" else f"render FPS, {args.workload}",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {W}x{H}, {N} Gaussians, {kind} covariance, "
                               f"model-init distribution, footprint x{fp}; one step = one batch "
                               f"of {B} different frames (seeds {1 + rank * B}..{rank * B + B}) "
                               "in one call",
                   "frames_per_step": B, "steps_in_flight": D, "intersections_per_frame": I_8d / B,
                   "list_entries_per_frame": I_held / B, "tiles_per_frame": T,
                   "parallelism": "1 process/GPU, independent frames per rank",
                   **({"collective_backend": dist.get_backend(),
                       "ranks_share_gpus": world > torch.cuda.device_count()}
                      if distributed else {})},
        "timed_regions": len(regions), "frames_timed": len(regions) * args.steps * B,
        "region_ms_min_median_max": [min(regions) * 1e3, sorted(regions)[len(regions) // 2] * 1e3,
                                     max(regions) * 1e3],
        "roofline": roofline,
    }

    # (the secondary figures are single-GPU figures: the one-rank run carries them)
    if not args.no_extras and rank == 0 and world == 1:
        result.update(secondary(args, device, batch, run, batch_in, B, I_8d, I_held,
                                timing["avg_launch_ms"]))
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        result["cpu_baseline"] = time_cpu_oracle(args.workload, budget_s=args.cpu_budget_s)
        result["cpu_baseline"]["gpu_over_cpu"] = fps / result["cpu_baseline"]["value"]

    if distributed:
        # Secondary measurement: one 4K frame split by tile rows.  It runs in a watched thread:
        # a collective that never completes (or a rank that died) must not cost the headline
        # line, so after GI2D_SHARD_TIMEOUT_S the ranks stop waiting, rank 0 prints what it has,
        # (its "tile_row_sharded" object then carries the error), and every rank leaves without
        # joining the stuck work.
        import threading
        box = {}

        def sharded():
            try:
                torch.cuda.set_device(local_rank)
                from gaussianimage_amd.sharding import bench_tile_row_sharded
                wl = "uhd200k"
                box["result"] = bench_tile_row_sharded(
                    device, rank, world, WORKLOADS[wl][:4], make_inputs(wl, device, seed=1),
                    steps=max(20, args.steps // 4), warmup=5, label=wl)
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
        # stuck or failed collective in the SECONDARY block: the headline line is out and says
        # so ("tile_row_sharded": {"error": ...}); do not wait for the stuck work at interpreter
        # exit.  GI2D_SHARD_STRICT=1 turns this into a non-zero exit status.
        os._exit(3 if os.environ.get("GI2D_SHARD_STRICT", "0") == "1" else 0)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
