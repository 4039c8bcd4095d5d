"""Builds the native library in-tree: ``gaussianimage_amd/csrc/libgi2d.so`` for gfx950."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["gi2d.hip"]
LIB = os.path.join(CSRC, "libgi2d.so")

# -ffp-contract=off: the projection's integer outputs must not depend on fused multiply-adds
#  chosen by the compiler; -packed-fp32-ops off: keep hipcc from forming v_pk_*_f32.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fno-fast-math",
               "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "gi2d.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, extra_flags=(), verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc(), *HIPCC_FLAGS, *extra_flags, "-o", LIB, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
