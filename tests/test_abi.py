"""CPU checks of the C-ABI boundary: the library loads without a GPU and exports
every symbol that include/gi2d.h declares; argument validation that needs no device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gi2d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gi2d_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from gaussianimage_amd import _lib
    names = _declared()
    assert len(names) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in gi2d.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_version_error_strings_and_defaults():
    from gaussianimage_amd import _lib
    assert _lib.lib.gi2d_version() == 100
    assert _lib.lib.gi2d_error_string(0) == b"ok"
    assert b"capacity" in _lib.lib.gi2d_error_string(_lib.GI2D_E_CAPACITY)
    c = _lib.default_conv()
    assert (c.pixel_offset, c.center_bias, c.alpha_cap, c.bg_weight) == (0.0, 0.0, 1.0, 0.0)
    assert abs(c.alpha_min - 1 / 255) < 1e-9


def test_null_arguments_are_rejected_without_a_device():
    from gaussianimage_amd import _lib
    lib = _lib.lib
    assert lib.gi2d_workspace_create(0, None) == _lib.GI2D_E_INVALID
    assert lib.gi2d_workspace_destroy(None) == _lib.GI2D_OK
    assert lib.gi2d_bin_tiles(None, None, None, 0, 1, 1, 0, None) == _lib.GI2D_E_INVALID
    conv = _lib.default_conv()
    # tile bounds that do not match the image are rejected before any launch
    rc = lib.gi2d_project_cholesky_fwd(None, None, 0, 32, 32, 3, 2, ctypes.byref(conv), None, None,
                                       None, None, None, None, None)
    assert rc == _lib.GI2D_E_INVALID


def test_operators_refuse_cpu_tensors():
    import torch
    from gsplat.project_gaussians_2d import project_gaussians_2d
    from gsplat.project_gaussians_2d_scale_rot import project_gaussians_2d_scale_rot
    from gsplat.rasterize_sum import rasterize_gaussians_sum
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        project_gaussians_2d(torch.zeros(4, 2), torch.ones(4, 3), 32, 32, (2, 2, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        project_gaussians_2d_scale_rot(torch.zeros(4, 2), torch.ones(4, 2), torch.zeros(4, 1), 32,
                                       32, (2, 2, 1))
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rasterize_gaussians_sum(z(4, 2), z(4), z(4, dtype=torch.int32), z(4, 3),
                                z(4, dtype=torch.int32), z(4, 3), z(4, 1), 32, 32, 16, 16,
                                background=torch.ones(3))
    # the reference wrappers' argument errors
    with pytest.raises(ValueError):
        rasterize_gaussians_sum(z(4, 2), z(4), z(4, dtype=torch.int32), z(4, 3),
                                z(4, dtype=torch.int32), z(4), z(4, 1), 32, 32, 16, 16)
    with pytest.raises(ValueError):
        rasterize_gaussians_sum(z(4, 2), z(4), z(4, dtype=torch.int32), z(4, 3),
                                z(4, dtype=torch.int32), z(4, 3), z(4), 32, 32, 16, 16)
    with pytest.raises(ValueError):
        rasterize_gaussians_sum(z(4, 2), z(4), z(4, dtype=torch.int32), z(4, 3),
                                z(4, dtype=torch.int32), z(4, 3), z(4, 1), 32, 32, 8, 8)


def test_library_is_built_from_the_sources_next_to_it():
    """The build id compiled into the library is the hash of the sources on disk (a stale
    library is refused at import, which this test would not even reach)."""
    from gaussianimage_amd import _lib, build
    assert _lib.lib.gi2d_build_id().decode() == build.source_id()
