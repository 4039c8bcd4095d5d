"""Decode path of the codec: the coded attributes of an image straight to pixels.

Mirrors ``decompress_wo_ec`` of the reference's models
(/root/reference/gaussianimage_cholesky.py:161-172, /root/reference/gaussianimage_rs.py:132-145,
timed by /root/reference/test_quantize.py:71-74): ``means = tanh(xyz.float())``, uniform
de-quantisation ``code * scale + beta`` (/root/reference/quantize.py:86-87) plus the model's
bound, colours as the sum of the residual-VQ layers' code vectors
(/root/reference/quantize.py:146-150), projection, rasterisation, clamp, permute.  Here the
decode is fused into the projection kernel and the clamp/permute into the rasteriser: two
launches per decoded frame, no intermediate tensors.

Producing the codes (k-means codebooks, learned quantiser scales, entropy coding) belongs to the
reference's training and bit-stream code and is not done here; ``synthetic_encoding`` makes
codes of the right shape and value ranges for tests and benchmarks.
"""
import ctypes

import torch

from . import _lib
from ._lib import BLOCK, check, lib
from .ops import Workspace, _ptr, _stream, get_conventions

__all__ = ["CodecDecoder", "synthetic_encoding"]


def _dev32(t, device, dtype=torch.float32):
    return t.detach().to(device=device, dtype=dtype).contiguous()


class CodecDecoder:
    """One coded image bound to pre-allocated device buffers.

    ``encoding`` is the reference's ``encoding_dict``: ``xyz`` (half, ``[N,2]``),
    ``feature_dc_index`` (``[N,Q]`` integer), and ``quant_cholesky_elements`` (``[N,3]``) for
    ``kind="cholesky"`` or ``quant_scaling`` (``[N,2]``) + ``quant_rotation`` (``[N,1]``) for
    ``kind="rs"``.  ``quantizers`` carries the models' quantiser parameters: ``cov_scale`` /
    ``cov_beta`` (``cholesky_quantizer`` or ``scaling_quantizer``), ``rot_scale`` / ``rot_beta``
    (``rotation_quantizer``, RS only) and ``codebooks`` ``[Q,K,3]`` (the layers'
    ``_codebook.embed[0]`` stacked).
    """

    def __init__(self, encoding, quantizers, H, W, kind="cholesky", device="cuda:0",
                 opacity=None, workspace=None):
        if kind not in ("cholesky", "rs"):
            raise ValueError("kind must be 'cholesky' or 'rs'")
        self.kind, self.H, self.W = kind, int(H), int(W)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CodecDecoder needs a GPU device (there is no CPU fallback)")
        d = self.device
        need = ["xyz", "feature_dc_index"] + (["quant_cholesky_elements"] if kind == "cholesky"
                                              else ["quant_scaling", "quant_rotation"])
        missing = [k for k in need if k not in encoding]
        if missing:
            raise ValueError("encoding lacks %s for kind=%r" % (missing, kind))
        xyz = encoding["xyz"]
        if xyz.dtype != torch.float16 or xyz.ndimension() != 2 or xyz.shape[1] != 2:
            raise ValueError("encoding['xyz'] must be a half tensor of shape (N, 2)")
        N = self.N = xyz.shape[0]
        self.xyz = xyz.detach().to(d).contiguous()
        idx = encoding["feature_dc_index"]
        if idx.ndimension() != 2 or idx.shape[0] != N:
            raise ValueError("feature_dc_index must have shape (N, Q)")
        self.index = _dev32(idx, d, torch.int32)
        cb = quantizers["codebooks"]
        if cb.ndimension() != 3 or cb.shape[0] != idx.shape[1] or cb.shape[2] != 3:
            raise ValueError("codebooks must have shape (Q, K, 3) with Q = feature_dc_index.shape[1]")
        if not 1 <= cb.shape[0] <= _lib.MAX_VQ_LAYERS:
            raise ValueError("1..%d residual-VQ layers" % _lib.MAX_VQ_LAYERS)
        self.codebooks = _dev32(cb, d)
        ca = 3 if kind == "cholesky" else 2
        code_a = encoding["quant_cholesky_elements" if kind == "cholesky" else "quant_scaling"]
        if tuple(code_a.shape) != (N, ca):
            raise ValueError("covariance codes must have shape (N, %d)" % ca)
        self.code_a = _dev32(code_a, d)
        self.scale_a = _dev32(quantizers["cov_scale"].reshape(-1), d)
        self.beta_a = _dev32(quantizers["cov_beta"].reshape(-1), d)
        if self.scale_a.numel() != ca or self.beta_a.numel() != ca:
            raise ValueError("cov_scale / cov_beta must have %d channels" % ca)
        self.code_b = self.scale_b = self.beta_b = None
        if kind == "rs":
            code_b = encoding["quant_rotation"]
            if code_b.numel() != N:
                raise ValueError("quant_rotation must have shape (N, 1)")
            self.code_b = _dev32(code_b.reshape(N, 1), d)
            self.scale_b = _dev32(quantizers["rot_scale"].reshape(-1), d)
            self.beta_b = _dev32(quantizers["rot_beta"].reshape(-1), d)
            if self.scale_b.numel() != 1 or self.beta_b.numel() != 1:
                raise ValueError("rot_scale / rot_beta must have 1 channel")
        self.codec = _lib.Codec(
            xyz_half=self.xyz.data_ptr(), cov_a_code=self.code_a.data_ptr(),
            cov_a_scale=self.scale_a.data_ptr(), cov_a_beta=self.beta_a.data_ptr(),
            cov_b_code=self.code_b.data_ptr() if kind == "rs" else None,
            cov_b_scale=self.scale_b.data_ptr() if kind == "rs" else None,
            cov_b_beta=self.beta_b.data_ptr() if kind == "rs" else None,
            color_index=self.index.data_ptr(), codebooks=self.codebooks.data_ptr(),
            num_quantizers=int(cb.shape[0]), codebook_size=int(cb.shape[1]))
        f32, i32 = torch.float32, torch.int32
        self.opacity = (torch.ones(N, 1, dtype=f32, device=d) if opacity is None
                        else _dev32(opacity, d))
        self.background = torch.ones(3, dtype=f32, device=d)
        self.xys = torch.empty(N, 2, dtype=f32, device=d)
        self.depths = torch.empty(N, dtype=f32, device=d)
        self.radii = torch.empty(N, dtype=i32, device=d)
        self.conics = torch.empty(N, 3, dtype=f32, device=d)
        self.num_tiles_hit = torch.empty(N, dtype=i32, device=d)
        self.colors = torch.empty(N, 3, dtype=f32, device=d)
        self.render = torch.empty(1, 3, self.H, self.W, dtype=f32, device=d)
        self.image = None
        self.ws = workspace or Workspace(d)
        self.conv = _lib.default_conv(**get_conventions())
        self.kind_id = 0 if kind == "cholesky" else 1
        self.tiles_y = (self.H + BLOCK - 1) // BLOCK

    def params(self):
        """The decoded operator inputs ``(means2d, cov_a, cov_b or None, colors)``."""
        d, N, f32 = self.device, self.N, torch.float32
        means = torch.empty(N, 2, dtype=f32, device=d)
        cov_a = torch.empty(N, 3 if self.kind == "cholesky" else 2, dtype=f32, device=d)
        cov_b = torch.empty(N, 1, dtype=f32, device=d) if self.kind == "rs" else None
        colors = torch.empty(N, 3, dtype=f32, device=d)
        check(lib.gi2d_decode_params(self.kind_id, ctypes.byref(self.codec), N, _ptr(means),
                                     _ptr(cov_a), _ptr(cov_b), _ptr(colors), _stream(d)),
              "gi2d_decode_params")
        return means, cov_a, cov_b, colors

    def decode(self, want_image=False):
        """``decompress_wo_ec(encoding_dict)``: returns ``{"render": [1,3,H,W]}`` (a buffer owned
        by the decoder, overwritten by the next call); ``want_image`` adds the unclamped
        ``[H,W,3]`` operator output under ``"image"``."""
        if want_image and self.image is None:
            self.image = torch.empty(self.H, self.W, 3, dtype=torch.float32, device=self.device)
        if self.ws.rows != (0, self.tiles_y):
            self.ws.set_rows(0, self.tiles_y)
        img = self.image if want_image else None
        check(lib.gi2d_decode_render(self.ws.handle, self.kind_id, ctypes.byref(self.codec),
                                     _ptr(self.opacity), _ptr(self.background), self.N, self.H,
                                     self.W, ctypes.byref(self.conv), _ptr(self.xys),
                                     _ptr(self.depths), _ptr(self.radii), _ptr(self.conics),
                                     _ptr(self.num_tiles_hit), _ptr(self.colors), _ptr(img),
                                     _ptr(self.render), _stream(self.device)),
              "gi2d_decode_render")
        self.ws.key = None
        out = {"render": self.render}
        if want_image:
            out["image"] = img
        return out


def synthetic_encoding(num_points, kind="cholesky", seed=0, footprint=1.0, bits=6,
                       num_quantizers=2, codebook_size=8):
    """Codes and quantiser parameters of the reference's shapes and ranges
    (/root/reference/gaussianimage_cholesky.py:47-49, gaussianimage_rs.py:39-42: 6-bit uniform
    quantisers, 2-layer residual VQ with 8 code vectors of dimension 3), drawn from the models'
    initial parameter distributions.  Returns ``(encoding, quantizers)`` of CPU tensors."""
    from ._testing import synth
    g = torch.Generator().manual_seed(seed)
    qmax = 2 ** bits - 1
    xyz = torch.atanh(2 * (torch.rand(num_points, 2, generator=g) - 0.5).clamp(-0.4999, 0.4999))
    enc = {"xyz": xyz.half(),
           "feature_dc_index": torch.randint(0, codebook_size, (num_points, num_quantizers),
                                             generator=g)}
    cb = torch.rand(num_quantizers, codebook_size, 3, generator=g)
    cb[1:] = (cb[1:] - 0.5) * 0.2  # residual layers: small corrections
    q = {"codebooks": cb}

    def quantise(x):  # UniformQuantizer._init_data + compress (quantize.py:44-49,82-84)
        if x.shape[0] == 0:
            return x.clone(), torch.ones(x.shape[1]), torch.zeros(x.shape[1])
        t_min, t_max = x.min(dim=0)[0], x.max(dim=0)[0]
        scale = (t_max - t_min) / qmax
        return ((x - t_min) / scale).clamp(0, qmax).round(), scale, t_min

    if kind == "cholesky":
        _, L, _, _ = synth.cholesky_inputs(num_points, seed=seed, footprint=footprint)
        raw = L - torch.tensor([0.5, 0.0, 0.5])
        enc["quant_cholesky_elements"], q["cov_scale"], q["cov_beta"] = quantise(raw)
    else:
        _, sc, rot, _, _ = synth.rs_inputs(num_points, seed=seed, footprint=footprint)
        enc["quant_scaling"], q["cov_scale"], q["cov_beta"] = quantise(sc - 0.5)
        enc["quant_rotation"], q["rot_scale"], q["rot_beta"] = quantise(rot.reshape(-1, 1))
    return enc, q
