"""Codec decode path (decompress_wo_ec): the oracle's restatement against the reference's torch
expressions on CPU, and the HIP path against the oracle on the GPU.

The reference holds no coded fixture for this path and its VQ container is a third-party
package that is not installed, so the codes are synthetic (reference shapes and ranges:
6-bit uniform codes, 2 residual-VQ layers of 8 code vectors)."""
import ctypes

import numpy as np
import pytest
import torch

from gaussianimage_amd import codec as codec_mod
from oracle import codec_oracle

DEV = "cuda:0"
KINDS = ["cholesky", "rs"]


def _np(t):
    return t.detach().cpu().numpy()


def _as_numpy(d):
    return {k: v.numpy() for k, v in d.items()}


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_follows_the_reference_expressions(kind):
    """gaussianimage_cholesky.py:163-166 / gaussianimage_rs.py:135-139 and quantize.py:86-87,
    146-150 written with torch on the CPU, line by line."""
    enc, q = codec_mod.synthetic_encoding(4000, kind, seed=3, footprint=2.0)
    means = torch.tanh(enc["xyz"].float())
    recon = 0
    for i in range(q["codebooks"].shape[0]):
        recon = recon + q["codebooks"][i][enc["feature_dc_index"][:, i]]
    got = codec_oracle.decode_params(kind, _as_numpy(enc), _as_numpy(q))
    # numpy's and torch's tanh are different vector implementations: a few ulp apart
    np.testing.assert_allclose(got[0], means.numpy(), rtol=0, atol=2.5e-7)
    assert np.array_equal(got[3], recon.numpy())
    if kind == "cholesky":
        chol = enc["quant_cholesky_elements"] * q["cov_scale"] + q["cov_beta"]
        chol = chol + torch.tensor([0.5, 0, 0.5]).view(1, 3)
        assert np.array_equal(got[1], chol.numpy()) and got[2] is None
    else:
        sc = torch.abs(enc["quant_scaling"] * q["cov_scale"] + q["cov_beta"]
                       + torch.tensor([0.5, 0.5]).view(1, 2))
        rot = enc["quant_rotation"] * q["rot_scale"] + q["rot_beta"]
        assert np.array_equal(got[1], sc.numpy())
        assert np.array_equal(got[2], rot.numpy())


@pytest.mark.parametrize("kind", KINDS)
def test_synthetic_codes_have_the_reference_ranges_and_round_trip(kind):
    from gaussianimage_amd._testing import synth
    N = 3000
    enc, q = codec_mod.synthetic_encoding(N, kind, seed=5)
    key = "quant_cholesky_elements" if kind == "cholesky" else "quant_scaling"
    codes = enc[key]
    assert codes.dtype == torch.float32 and codes.min() >= 0 and codes.max() <= 63
    assert torch.equal(codes, codes.round())
    assert enc["xyz"].dtype == torch.float16 and enc["feature_dc_index"].max() < 8
    got = codec_oracle.decode_params(kind, _as_numpy(enc), _as_numpy(q))
    if kind == "cholesky":
        want = synth.cholesky_inputs(N, seed=5)[1].numpy()
    else:
        want = synth.rs_inputs(N, seed=5)[1].numpy()
    # de-quantisation lands within half a step of the value that was coded
    assert np.abs(got[1] - want).max() <= 0.5 * float(q["cov_scale"].max()) + 1e-6


def test_decode_entry_points_reject_bad_arguments_without_a_device():
    from gaussianimage_amd import _lib
    cd = _lib.Codec()
    assert _lib.lib.gi2d_decode_params(0, None, 4, None, None, None, None, None) == _lib.GI2D_E_INVALID
    cd.num_quantizers, cd.codebook_size = 2, 8
    assert _lib.lib.gi2d_decode_params(0, ctypes.byref(cd), 4, None, None, None, None,
                                       None) == _lib.GI2D_E_INVALID          # NULL streams of codes
    assert _lib.lib.gi2d_decode_params(2, ctypes.byref(cd), 0, None, None, None, None,
                                       None) == _lib.GI2D_E_INVALID          # unknown kind
    cd.num_quantizers = 99
    assert _lib.lib.gi2d_decode_params(0, ctypes.byref(cd), 0, None, None, None, None,
                                       None) == _lib.GI2D_E_INVALID
    conv = _lib.default_conv()
    assert _lib.lib.gi2d_decode_render(None, 0, ctypes.byref(cd), None, None, 0, 16, 16,
                                       ctypes.byref(conv), None, None, None, None, None, None,
                                       None, None, None) == _lib.GI2D_E_INVALID
    enc, q = codec_mod.synthetic_encoding(10)
    with pytest.raises(RuntimeError):
        codec_mod.CodecDecoder(enc, q, 32, 32, device="cpu")


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("H,W,N,seed,fp", [(48, 64, 64, 0, 1.0), (100, 130, 2000, 3, 3.0),
                                           (512, 768, 30000, 1, 1.0)])
def test_decoded_parameters_match_the_oracle(kind, H, W, N, seed, fp):
    enc, q = codec_mod.synthetic_encoding(N, kind, seed=seed, footprint=fp)
    dec = codec_mod.CodecDecoder(enc, q, H, W, kind=kind, device=DEV)
    means, cov_a, cov_b, colors = dec.params()
    ref = codec_oracle.decode_params(kind, _as_numpy(enc), _as_numpy(q))
    assert np.array_equal(_np(cov_a), ref[1]), "covariance factors: bit for bit"
    assert np.array_equal(_np(colors), ref[3]), "colours: bit for bit"
    if kind == "rs":
        assert np.array_equal(_np(cov_b), ref[2]), "rotation: bit for bit"
    # tanh of the device and of numpy: a few ulp
    np.testing.assert_allclose(_np(means), ref[0], rtol=0, atol=2.5e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("H,W,N,seed,fp", [(48, 64, 64, 0, 1.0), (100, 130, 2000, 3, 3.0),
                                           (512, 768, 30000, 1, 1.0)])
def test_decode_render_equals_operator_path_and_oracle(oracle, kind, H, W, N, seed, fp):
    """Two-launch decode == decoded parameters pushed through the drop-in operators with the
    reference's clamp/view/permute (bit for bit), and == the oracle's rasterisation."""
    from gaussianimage_amd import ops
    enc, q = codec_mod.synthetic_encoding(N, kind, seed=seed, footprint=fp)
    dec = codec_mod.CodecDecoder(enc, q, H, W, kind=kind, device=DEV)
    out = dec.decode(want_image=True)
    render, image = out["render"].clone(), out["image"].clone()
    assert render.shape == (1, 3, H, W) and image.shape == (H, W, 3)
    means, cov_a, cov_b, colors = dec.params()
    tb = oracle.tile_bounds(H, W)
    if kind == "cholesky":
        proj = ops.project_gaussians_2d(means, cov_a, H, W, tb)
    else:
        proj = ops.project_gaussians_2d_scale_rot(means, cov_a, cov_b, H, W, tb)
    img2 = ops.rasterize_gaussians_sum(*proj, colors, dec.opacity, H, W, 16, 16,
                                       background=dec.background, return_alpha=False)
    ref_render = torch.clamp(img2, 0, 1).view(-1, H, W, 3).permute(0, 3, 1, 2).contiguous()
    assert torch.equal(image, img2)
    assert torch.equal(render, ref_render)
    for got, want in zip((dec.xys, dec.radii, dec.conics, dec.num_tiles_hit, dec.colors),
                         (proj[0], proj[2], proj[3], proj[4], colors)):
        assert torch.equal(got, want)
    # against the oracle: parameters from the oracle's decode, geometry as the device rounded it
    ref = codec_oracle.decode_params(kind, _as_numpy(enc), _as_numpy(q))
    if kind == "cholesky":
        r_proj = oracle.project_cholesky_fwd(ref[0], ref[1], H, W)
        assert np.array_equal(_np(dec.radii), r_proj[2])
        assert np.array_equal(_np(dec.conics), r_proj[3])
        np.testing.assert_allclose(_np(dec.xys), r_proj[0], rtol=0, atol=1e-4)
    offs, ids = oracle.bin_tiles(_np(dec.xys), _np(dec.radii), H, W)
    r_img, margin = oracle.rasterize_sum_fwd(_np(dec.xys), _np(dec.conics), ref[3],
                                             _np(dec.opacity), H, W, offs, ids, want_margin=True)
    ok = margin > 1e-4
    assert ok.mean() > 0.99
    scale = max(1.0, float(np.abs(r_img).max()))
    assert np.abs(_np(image) - r_img)[ok].max() <= 1e-5 * scale
    r_render = np.clip(r_img, 0, 1).reshape(1, H, W, 3).transpose(0, 3, 1, 2)
    ok_chw = np.broadcast_to(ok.reshape(1, 1, H, W), (1, 3, H, W))
    assert np.abs(_np(render) - r_render)[ok_chw].max() <= 1e-5 * scale
    # a second decode reuses every buffer and gives the same bits
    again = dec.decode()["render"]
    assert torch.equal(again, render)


@pytest.mark.gpu
def test_decode_edge_cases():
    # no Gaussians: a black frame
    enc, q = codec_mod.synthetic_encoding(0)
    dec = codec_mod.CodecDecoder(enc, q, 40, 50, device=DEV)
    out = dec.decode(want_image=True)
    assert not out["render"].any() and not out["image"].any()
    # an index outside the codebook is clamped into it
    enc, q = codec_mod.synthetic_encoding(100, seed=2)
    bad = {k: v.clone() for k, v in enc.items()}
    bad["feature_dc_index"][:, 0] = 1000
    bad["feature_dc_index"][:, 1] = -5
    top = {k: v.clone() for k, v in enc.items()}
    top["feature_dc_index"][:, 0] = 7
    top["feature_dc_index"][:, 1] = 0
    a = codec_mod.CodecDecoder(bad, q, 64, 64, device=DEV).params()[3]
    b = codec_mod.CodecDecoder(top, q, 64, 64, device=DEV).params()[3]
    assert torch.equal(a, b)
    # argument checks of the host side
    with pytest.raises(ValueError):
        codec_mod.CodecDecoder({**enc, "xyz": enc["xyz"].float()}, q, 64, 64, device=DEV)
    with pytest.raises(ValueError):
        codec_mod.CodecDecoder(enc, {**q, "codebooks": q["codebooks"][:1]}, 64, 64, device=DEV)
    with pytest.raises(ValueError):
        codec_mod.CodecDecoder(enc, q, 64, 64, kind="rs", device=DEV)
