"""The two functions the harness calls: what build() leaves behind must be in the tree and
current, smoke() must pass on the GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_libraries_are_built_in_tree_and_current():
    """(build() itself is not called here: it rewrites libraries this process has loaded)"""
    import __graft_entry__ as entry
    from gaussianimage_amd import build
    assert callable(entry.build) and callable(entry.smoke)
    assert os.path.exists(build.LIB) and not build.stale()
    assert os.path.exists(os.path.join(ROOT, "oracle", "libgi2d_oracle.so"))


@pytest.mark.gpu
def test_smoke_passes():
    import __graft_entry__ as entry
    entry.smoke()
