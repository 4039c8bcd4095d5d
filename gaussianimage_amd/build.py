"""Builds the native library in-tree: ``gaussianimage_amd/csrc/libgi2d.so`` for gfx950.

One hipcc compile per translation unit (in parallel), one link.  Objects go to
``csrc/_build`` (ignored by git); only the ``.so`` is loaded at run time."""
import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["lists.hip", "project.hip", "raster_fwd.hip", "raster_bwd.hip", "fit.hip",
           "noalloc.hip"]
LIB = os.path.join(CSRC, "libgi2d.so")
OBJ_DIR = os.path.join(CSRC, "_build")

# -ffp-contract=off: the projection's integer outputs must not depend on fused multiply-adds
#  chosen by the compiler; -packed-fp32-ops off: keep hipcc from forming v_pk_*_f32;
#  hidden visibility on the host side: the library exports its C ABI and nothing else.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-ffp-contract=off", "-fno-fast-math",
               "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
               "-Xarch_host", "-fvisibility=hidden"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _deps():
    deps = [os.path.join(CSRC, s) for s in SOURCES] + glob.glob(os.path.join(CSRC, "*.h"))
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "gi2d.h"))
    return deps


def source_id():
    """Hash of every source the library is built from (the .hip / .h files of csrc and the
    public header): compiled into the library as gi2d_build_id() and compared at load time, so
    that a library older than its sources cannot be used unnoticed."""
    h = hashlib.sha256()
    for d in sorted(_deps()):
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_native(force=False, extra_flags=(), verbose=False, jobs=None):
    if not force and not stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    cmds = []
    build_id = '-DGI2D_BUILD_ID="%s"' % source_id()
    for s in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        cmds.append([hipcc(), *HIPCC_FLAGS, *extra_flags, build_id, "-c", os.path.join(CSRC, s),
                     "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=jobs or min(len(cmds), os.cpu_count() or 1, 8)) as pool:
        list(pool.map(run, cmds))
    run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
