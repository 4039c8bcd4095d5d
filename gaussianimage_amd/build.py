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


TORCH_OPS_SRC = os.path.join(CSRC, "torch_ops.cpp")
TORCH_OPS_LIB = os.path.join(_HERE, "_torch_ops.so")


def torch_ops_stale():
    if not os.path.exists(TORCH_OPS_LIB):
        return True
    t = os.path.getmtime(TORCH_OPS_LIB)
    deps = [TORCH_OPS_SRC, os.path.join(os.path.dirname(_HERE), "include", "gi2d.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def torch_ops_id():
    """Hash of what the compiled binding is built from, and the torch it is built against."""
    import torch
    h = hashlib.sha256()
    for d in (TORCH_OPS_SRC, os.path.join(os.path.dirname(_HERE), "include", "gi2d.h")):
        h.update(open(d, "rb").read())
    return h.hexdigest()[:16] + "/torch-" + torch.__version__


def build_torch_ops(force=False, verbose=False):
    """The compiled binding of the three operators (csrc/torch_ops.cpp -> _torch_ops.so): host
    code only, compiled with g++ against the installed torch's headers and linked to libgi2d.so
    (found at run time next to it: rpath $ORIGIN/csrc).  No device code, no JIT cache."""
    if not force and not torch_ops_stale():
        return TORCH_OPS_LIB
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_torch_ops", "-DTORCH_API_INCLUDE_EXTENSION_H",
           '-DGI2D_TORCH_OPS_ID="%s"' % torch_ops_id(),
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-attributes"]
    for inc in ce.include_paths():
        cmd += ["-isystem", inc]
    cmd += ["-isystem", "/opt/rocm/include", "-I", sysconfig.get_paths()["include"],
            "-I", os.path.join(os.path.dirname(_HERE), "include"), TORCH_OPS_SRC, "-o",
            TORCH_OPS_LIB]
    for lib_dir in ce.library_paths():
        cmd += ["-L", lib_dir]
    cmd += ["-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-L", CSRC, "-lgi2d", "-Wl,-rpath,$ORIGIN/csrc"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_OPS_LIB


C_HOST_SRC = os.path.join(os.path.dirname(_HERE), "examples", "c_host", "render_batch.c")
C_HOST_BIN = os.path.join(os.path.dirname(_HERE), "examples", "c_host", "render_batch")


def build_examples(force=False, verbose=False):
    """The C host of examples/c_host (plain C99 through gcc: the C ABI needs no C++ and no HIP
    compiler on the caller's side), linked to libgi2d.so and the HIP runtime."""
    deps = [C_HOST_SRC, os.path.join(os.path.dirname(_HERE), "include", "gi2d.h"), LIB]
    if not force and os.path.exists(C_HOST_BIN) and \
            all(os.path.getmtime(d) <= os.path.getmtime(C_HOST_BIN) for d in deps):
        return C_HOST_BIN
    cmd = [shutil.which("gcc") or "gcc", "-O2", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__",
           "-I", "/opt/rocm/include", "-I", os.path.join(os.path.dirname(_HERE), "include"),
           C_HOST_SRC, "-o", C_HOST_BIN, "-L", CSRC, "-lgi2d", "-L", "/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../gaussianimage_amd/csrc",
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return C_HOST_BIN


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
    print(build_torch_ops(force=True, verbose=True))
    print(build_examples(force=True, verbose=True))
