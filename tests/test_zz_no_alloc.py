"""gi2d_render_frame_noalloc -- the frame call whose tile lists live in a file-scope device
array -- against gi2d_render_frame with a workspace, through the C ABI: the same bits, on two
calls in a row (the second with a smaller input, so that what the first left in the shared
array must not leak), with no change in free device memory across the calls, and loud limits.

Collected last on purpose (file name): it is about an entry point beside the hot path."""
import ctypes

import numpy as np
import pytest
import torch

from gaussianimage_amd._testing import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _frame(lib_mod, kind, N, H, W, seed, footprint, ws_handle):
    """Buffers + a filled gi2d_frame_args; ws_handle None = for the no-alloc call."""
    from gaussianimage_amd.ops import _ptr
    f32, i32 = torch.float32, torch.int32
    if kind == 0:
        means, a, col, op = synth.cholesky_inputs(N, seed=seed, footprint=footprint)
        b = None
    else:
        means, a, b, col, op = synth.rs_inputs(N, seed=seed, footprint=footprint)
    ins = [None if t is None else t.to(DEV).contiguous() for t in (means, a, b, col)]
    ins.append(op.reshape(-1).to(DEV))
    outs = dict(xys=torch.empty(N, 2, dtype=f32, device=DEV),
                depths=torch.empty(N, dtype=f32, device=DEV),
                radii=torch.empty(N, dtype=i32, device=DEV),
                conics=torch.empty(N, 3, dtype=f32, device=DEV),
                num_tiles_hit=torch.empty(N, dtype=i32, device=DEV),
                out_img=torch.full((H, W, 3), -7.0, dtype=f32, device=DEV))
    args = lib_mod.FrameArgs(
        ws=ws_handle, kind=kind, means2d=_ptr(ins[0]), cov_a=_ptr(ins[1]), cov_b=_ptr(ins[2]),
        colors=_ptr(ins[3]), opacity=_ptr(ins[4]), background=None, N=N, H=H, W=W,
        conv=lib_mod.default_conv(), raster_begin_event=None, raster_end_event=None,
        stream=torch.cuda.current_stream().cuda_stream or None,
        **{k: _ptr(v) for k, v in outs.items()})
    return args, outs, ins


CASES = [(0, 3000, 96, 128, 1, 2.0), (1, 500, 48, 64, 2, 1.0), (0, 30000, 512, 768, 1, 1.0),
         (0, 40, 33, 17, 3, 1.0)]


def test_noalloc_frame_equals_the_workspace_frame():
    from gaussianimage_amd import _lib
    from gaussianimage_amd.ops import Workspace
    lib = _lib.lib
    # reference results first (these calls allocate: workspaces, lists)
    want = []
    for kind, N, H, W, seed, fp in CASES:
        ws = Workspace(DEV)
        args, outs, keep = _frame(_lib, kind, N, H, W, seed, fp, ws.handle)
        _lib.check(lib.gi2d_render_frame(ctypes.byref(args)), "gi2d_render_frame")
        torch.cuda.synchronize()
        want.append({k: v.clone() for k, v in outs.items()})
        del ws
    # the no-alloc calls, back to back on one stream: big, small, bigger, tiny
    frames = [_frame(_lib, kind, N, H, W, seed, fp, None) for kind, N, H, W, seed, fp in CASES]
    _lib.check(lib.gi2d_render_frame_noalloc(ctypes.byref(frames[0][0]), 0), "warm-up")
    torch.cuda.synchronize()
    frames[0][1]["out_img"].fill_(-7.0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for (args, outs, keep), cap in zip(frames, (0, 512, 256, 0)):
        _lib.check(lib.gi2d_render_frame_noalloc(ctypes.byref(args), cap),
                   "gi2d_render_frame_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "the calls changed the device's free memory"
    for (args, outs, keep), w, case in zip(frames, want, CASES):
        for k in outs:
            assert torch.equal(outs[k], w[k]), (case, k)
        assert torch.isfinite(outs["out_img"]).all()
    # ... and once more in the opposite order (the array keeps the last call's contents)
    for (args, outs, keep) in reversed(frames):
        outs["out_img"].fill_(-7.0)
        _lib.check(lib.gi2d_render_frame_noalloc(ctypes.byref(args), 0),
                   "gi2d_render_frame_noalloc")
    torch.cuda.synchronize()
    for (args, outs, keep), w, case in zip(frames, want, CASES):
        assert torch.equal(outs["out_img"], w["out_img"]), case


def test_noalloc_limits_are_loud():
    from gaussianimage_amd import _lib
    from gaussianimage_amd.ops import Workspace
    lib = _lib.lib
    args, outs, keep = _frame(_lib, 0, 100, 64, 64, 1, 1.0, None)
    # more tiles than the array holds
    big = _lib.FrameArgs.from_buffer_copy(args)
    big.H, big.W = 16 * 200, 16 * 200
    assert lib.gi2d_render_frame_noalloc(ctypes.byref(big), 0) == _lib.GI2D_E_CAPACITY
    # more ids than the array holds: 16 tiles x 200 000 ids
    assert lib.gi2d_render_frame_noalloc(ctypes.byref(args), 200000) == _lib.GI2D_E_CAPACITY
    assert lib.gi2d_render_frame_noalloc(ctypes.byref(args), -1) == _lib.GI2D_E_INVALID
    ws = Workspace(DEV)
    with_ws = _lib.FrameArgs.from_buffer_copy(args)
    with_ws.ws = ws.handle
    assert lib.gi2d_render_frame_noalloc(ctypes.byref(with_ws), 0) == _lib.GI2D_E_INVALID
    assert lib.gi2d_render_frame_noalloc(None, 0) == _lib.GI2D_E_INVALID
    # a list longer than the capacity: that tile is NaN, the others are right
    N, H, W = 4000, 32, 32
    args, outs, keep = _frame(_lib, 0, N, H, W, 4, 3.0, None)
    _lib.check(lib.gi2d_render_frame_noalloc(ctypes.byref(args), 64), "small capacity")
    torch.cuda.synchronize()
    assert torch.isnan(outs["out_img"]).any()
    _lib.check(lib.gi2d_render_frame_noalloc(ctypes.byref(args), 0), "full capacity")
    torch.cuda.synchronize()
    assert torch.isfinite(outs["out_img"]).all()
