"""The compiled binding of the three operators (csrc/torch_ops.cpp) against the ctypes binding
(ops.py): same kernels, same workspace policy, so the same BITS -- outputs, images, gradients --
and the same errors.  The CPU part checks that it builds, loads and rejects bad arguments."""
import math

import pytest
import torch

from gaussianimage_amd import build, ops
from gaussianimage_amd._testing import synth

DEV = "cuda:0"


def test_compiled_binding_is_built_from_the_current_sources():
    assert ops._C is not None, ops._C_WHY_NOT
    assert ops._C.build_id() == build.torch_ops_id()
    assert ops._C.abi_version() == 100
    assert ops.compiled_binding()


def test_compiled_binding_argument_errors_without_a_gpu():
    m, L = torch.zeros(4, 2), torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.project_gaussians_2d(m, L, 32, 32, (2, 2, 1))
    with pytest.raises(ValueError, match="means2d must have shape"):
        ops.project_gaussians_2d(L, L, 32, 32, (2, 2, 1))
    with pytest.raises(ValueError, match="L_elements must have shape"):
        ops.project_gaussians_2d(m, m, 32, 32, (2, 2, 1))
    with pytest.raises(ValueError, match="scales2d must have shape"):
        ops.project_gaussians_2d_scale_rot(m, L, torch.zeros(4, 1), 32, 32, (2, 2, 1))
    with pytest.raises(ValueError, match="rotation must have shape"):
        ops.project_gaussians_2d_scale_rot(m, m, torch.zeros(3, 1), 32, 32, (2, 2, 1))
    z, zi = torch.zeros(4), torch.zeros(4, dtype=torch.int32)
    args = (m, z, zi, L, zi)
    with pytest.raises(ValueError, match="colors must have dimensions"):
        ops.rasterize_gaussians_sum(*args, torch.zeros(4), torch.zeros(4, 1), 32, 32, 16, 16)
    with pytest.raises(ValueError, match="opacity must have dimensions"):
        ops.rasterize_gaussians_sum(*args, L, torch.zeros(4), 32, 32, 16, 16)
    with pytest.raises(ValueError, match="same number of channels"):
        ops.rasterize_gaussians_sum(*args, torch.zeros(4, 4), torch.zeros(4, 1), 32, 32, 16, 16,
                                    background=torch.ones(3))
    with pytest.raises(ValueError, match="tile size is fixed"):
        ops.rasterize_gaussians_sum(*args, L, torch.zeros(4, 1), 32, 32, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rasterize_gaussians_sum(*args, L, torch.zeros(4, 1), 32, 32, 16, 16)


def test_binding_switch():
    prev = ops.use_compiled(False)
    try:
        assert not ops.compiled_binding()
        with pytest.raises(ValueError, match="means2d must have shape"):
            ops.project_gaussians_2d(torch.zeros(4, 3), torch.zeros(4, 3), 32, 32, (2, 2, 1))
    finally:
        ops.use_compiled(prev)
    assert ops.compiled_binding() == prev


# ---------------------------------------------------------------------------------------------
def _cholesky_run(H, W, N, seed, passes, v_seed):
    """One reference-style forward (three raster passes sharing a binning) + backward."""
    means, L, col, op = synth.cholesky_inputs(N, seed=seed)
    m, Lc, c = (t.to(DEV).requires_grad_(True) for t in (means, L, col))
    o = op.to(DEV).requires_grad_(True)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    proj = ops.project_gaussians_2d(m, Lc, H, W, tb)
    bg = torch.tensor([0.2, 0.5, 0.9], device=DEV)
    imgs = [ops.rasterize_gaussians_sum(*proj, c, o, H, W, 16, 16, background=bg)]
    if passes == 3:
        imgs.append(ops.rasterize_gaussians_sum(*proj, (c * 0.5).detach(), o.detach(), H, W, 16, 16,
                                                background=bg, return_alpha=False))
        imgs.append(ops.rasterize_gaussians_sum(*proj, torch.ones_like(c), o.detach(), H, W, 16,
                                                16))
    g = torch.Generator().manual_seed(v_seed)
    v = torch.randn(H, W, 3, generator=g).to(DEV)
    # a non-contiguous incoming gradient, as the reference's permute/contiguous produces
    out = imgs[0].view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()
    (out * v.permute(2, 0, 1)[None]).sum().backward()
    return ([t.detach().clone() for t in proj] + [i.detach().clone() for i in imgs] +
            [t.grad.clone() for t in (m, Lc, c, o)])


def _rs_run(H, W, N, seed):
    means, sc, rot, col, op = synth.rs_inputs(N, seed=seed)
    m, s, r, c = (t.to(DEV).requires_grad_(True) for t in (means, sc, rot, col))
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    proj = ops.project_gaussians_2d_scale_rot(m, s, r, H, W, tb)
    img = ops.rasterize_gaussians_sum(*proj, c, op.to(DEV), H, W, 16, 16)
    torch.nn.functional.mse_loss(torch.clamp(img, 0, 1), torch.full_like(img, 0.3)).backward()
    return ([t.detach().clone() for t in proj] + [img.detach().clone()] +
            [t.grad.clone() for t in (m, s, r, c)])


def _both(fn, *args):
    prev = ops.use_compiled(True)
    try:
        a = fn(*args)
        ops.use_compiled(False)
        b = fn(*args)
    finally:
        ops.use_compiled(prev)
    return a, b


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(96, 128, 1500), (512, 768, 30000), (200, 176, 5000)])
@pytest.mark.parametrize("passes", [1, 3])
def test_cholesky_model_step_bits_match_the_ctypes_binding(shape, passes):
    H, W, N = shape
    a, b = _both(_cholesky_run, H, W, N, 3, passes, 11)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and x.shape == y.shape, i
        assert torch.equal(x, y), i


@pytest.mark.gpu
def test_rs_model_step_bits_match_the_ctypes_binding():
    a, b = _both(_rs_run, 512, 768, 30000, 2)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and x.shape == y.shape, i
        assert torch.equal(x, y), i


@pytest.mark.gpu
def test_no_grad_forward_and_input_conversions():
    H, W, N = 96, 128, 1200
    means, L, col, op = synth.cholesky_inputs(N, seed=5)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)

    def run():
        with torch.no_grad():
            # float64 and non-contiguous inputs are converted, as the ctypes binding does
            m = means.to(DEV).double()
            Lw = torch.zeros(N, 6, device=DEV)
            Lw[:, ::2] = L.to(DEV)
            proj = ops.project_gaussians_2d(m, Lw[:, ::2], H, W, tb)
            assert not proj[0].requires_grad and proj[2].dtype == torch.int32
            img = ops.rasterize_gaussians_sum(proj[0], proj[1], proj[2].long(), proj[3], proj[4],
                                              col.to(DEV).double(), op.to(DEV), H, W, 16, 16)
            return list(proj) + [img]
    a, b = _both(run)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_one_binning_for_the_three_passes_and_rebinning_after_an_edit():
    H, W, N = 96, 128, 1200
    means, L, col, op = synth.cholesky_inputs(N, seed=6)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)
    prev = ops.use_compiled(True)
    try:
        ws = ops.get_workspace(DEV)
        with torch.no_grad():
            proj = ops.project_gaussians_2d(means.to(DEV), L.to(DEV), H, W, tb)
            gen0 = ws.info()["generation"]
            a = ops.rasterize_gaussians_sum(*proj, col.to(DEV), op.to(DEV), H, W, 16, 16)
            gen1 = ws.info()["generation"]
            b = ops.rasterize_gaussians_sum(*proj, col.to(DEV), op.to(DEV), H, W, 16, 16)
            assert ws.info()["generation"] == gen1 and torch.equal(a, b)
            assert gen1 >= gen0
            # an in-place edit of xys bumps its version: the lists are rebuilt from the new values
            proj[0].add_(8.0)
            c = ops.rasterize_gaussians_sum(*proj, col.to(DEV), op.to(DEV), H, W, 16, 16)
            ops.use_compiled(False)
            want = ops.rasterize_gaussians_sum(*proj, col.to(DEV), op.to(DEV), H, W, 16, 16)
            assert torch.equal(c, want) and not torch.equal(c, a)
    finally:
        ops.use_compiled(prev)


@pytest.mark.gpu
def test_two_backward_passes_over_one_forward():
    H, W, N = 96, 128, 1500
    means, L, col, op = synth.cholesky_inputs(N, seed=7)
    tb = ((W + 15) // 16, (H + 15) // 16, 1)

    def run():
        m, Lc, c = (t.to(DEV).requires_grad_(True) for t in (means, L, col))
        proj = ops.project_gaussians_2d(m, Lc, H, W, tb)
        img = ops.rasterize_gaussians_sum(*proj, c, op.to(DEV), H, W, 16, 16)
        g1 = torch.autograd.grad(img.sum(), (m, Lc, c), retain_graph=True)
        g2 = torch.autograd.grad((img * img).sum(), (m, Lc, c))
        return list(g1) + list(g2)
    a, b = _both(run)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not torch.equal(a[0], a[3])


@pytest.mark.gpu
def test_status_errors_surface_as_gi2d_errors():
    from gaussianimage_amd import _lib
    prev = ops.use_compiled(True)
    try:
        N = 8
        xys = torch.rand(N, 2, device=DEV)
        with pytest.raises(_lib.Gi2dError) as e:
            # an image too large for the library's tile grid is refused with a status
            ops.rasterize_gaussians_sum(xys, torch.zeros(N, device=DEV),
                                        torch.ones(N, dtype=torch.int32, device=DEV),
                                        torch.rand(N, 3, device=DEV),
                                        torch.zeros(N, dtype=torch.int32, device=DEV),
                                        torch.rand(N, 3, device=DEV), torch.ones(N, 1, device=DEV),
                                        0, 64, 16, 16)
        assert e.value.status < 0
    finally:
        ops.use_compiled(prev)
