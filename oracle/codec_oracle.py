"""CPU restatement of the reference's codec decode path -- TEST INFRASTRUCTURE ONLY.

Only tests, ``__graft_entry__.smoke()`` and the benchmark's CPU baseline may import this file;
the product (``gaussianimage_amd``) never does.

Follows, step by step and in float32 with one rounding per operation:
  * /root/reference/gaussianimage_cholesky.py:161-166 -- ``decompress_wo_ec`` of the Cholesky
    model: ``means = tanh(xyz.float())``, ``cholesky = decompress(codes) + cholesky_bound``;
  * /root/reference/gaussianimage_rs.py:132-139 -- the RS model:
    ``scaling = |decompress(codes) + bound|``, ``rotation = decompress(codes)``;
  * /root/reference/quantize.py:86-87 -- ``UniformQuantizer.decompress``: ``x * scale + beta``;
  * /root/reference/quantize.py:146-150 -- ``VectorQuantizer.decompress``: the sum over the
    residual-VQ layers of ``layer._codebook.embed[0, index[:, i]]``.

PARITY UNPINNED for the VQ layer container: ``vector_quantize_pytorch`` (the reference's
requirements.txt, unpinned version) is not installed, so the layout ``embed[0]`` = ``[K, dim]``
is taken from the call site above, and the reference ships no coded fixture for this path.
The arithmetic itself is fully determined by the cited lines.
"""
import numpy as np

CHOLESKY_BOUND = np.array([0.5, 0.0, 0.5], np.float32)  # gaussianimage_cholesky.py:42
RS_BOUND = np.array([0.5, 0.5], np.float32)             # gaussianimage_rs.py:35


def uniform_decompress(codes, scale, beta):
    """quantize.py:86-87."""
    codes = np.asarray(codes, np.float32)
    return (codes * np.asarray(scale, np.float32).reshape(1, -1)).astype(np.float32) + \
        np.asarray(beta, np.float32).reshape(1, -1)


def vq_decompress(index, codebooks):
    """quantize.py:146-150; ``codebooks`` is ``[Q, K, dim]`` (each layer's ``embed[0]``)."""
    index = np.asarray(index)
    codebooks = np.asarray(codebooks, np.float32)
    recon = np.zeros((index.shape[0], codebooks.shape[2]), np.float32)
    for i in range(codebooks.shape[0]):
        recon = recon + codebooks[i][index[:, i]]
    return recon


def decode_params(kind, encoding, quantizers):
    """The operator inputs ``(means2d, cov_a, cov_b or None, colors)`` of decompress_wo_ec."""
    xyz = np.asarray(encoding["xyz"])
    assert xyz.dtype == np.float16
    means = np.tanh(xyz.astype(np.float32)).astype(np.float32)
    colors = vq_decompress(encoding["feature_dc_index"], quantizers["codebooks"])
    if kind == "cholesky":
        chol = uniform_decompress(encoding["quant_cholesky_elements"], quantizers["cov_scale"],
                                  quantizers["cov_beta"])
        return means, (chol + CHOLESKY_BOUND.reshape(1, 3)).astype(np.float32), None, colors
    scaling = uniform_decompress(encoding["quant_scaling"], quantizers["cov_scale"],
                                 quantizers["cov_beta"])
    scaling = np.abs(scaling + RS_BOUND.reshape(1, 2)).astype(np.float32)
    rotation = uniform_decompress(np.asarray(encoding["quant_rotation"]).reshape(-1, 1),
                                  quantizers["rot_scale"], quantizers["rot_beta"])
    return means, scaling, rotation.astype(np.float32), colors


def decode_render(kind, encoding, quantizers, H, W, port, opacity=None):
    """decompress_wo_ec as a whole; ``port`` is the C oracle front end (oracle.cport).  Returns
    ``(render[1,3,H,W], image[H,W,3], margin)``: the clamp and permute of
    gaussianimage_cholesky.py:170-171 applied to the oracle's rasterisation."""
    means, cov_a, cov_b, colors = decode_params(kind, encoding, quantizers)
    if kind == "cholesky":
        xys, depths, radii, conics, nth = port.project_cholesky_fwd(means, cov_a, H, W)
    else:
        xys, depths, radii, conics, nth = port.project_rs_fwd(means, cov_a, cov_b, H, W)
    offs, ids = port.bin_tiles(xys, radii, H, W)
    op = np.ones((means.shape[0], 1), np.float32) if opacity is None else opacity
    img, margin = port.rasterize_sum_fwd(xys, conics, colors, op, H, W, offs, ids,
                                         want_margin=True)
    render = np.clip(img, 0.0, 1.0).reshape(1, H, W, 3).transpose(0, 3, 1, 2).copy()
    return render, img, margin
