"""Guards on the machine code that actually ships: the gfx950 code objects inside
``csrc/libgi2d.so`` are pulled out of the library (no GPU needed) and disassembled.

* no packed-FP32 arithmetic (``v_pk_add_f32`` / ``v_pk_mul_f32`` / ``v_pk_fma_f32``): the build
  passes ``-target-feature -packed-fp32-ops`` to the device compiler -- the host pass answers that
  it does not know the feature, the device pass honours it, and this test is what shows it
  (without the flag hipcc -O3 forms over a hundred of them in the rasterisers);
* no scratch (private-memory) traffic: nothing spills;
* no writes through the scalar data cache (scalar stores, scalar atomics, its write-back);
* no fused multiply-add on doubles in the projection (the shared sine / cosine must stay a
  sequence of separately rounded operations).

This file only names those instructions in order to look for them; it is listed in
``.gpurunignore`` and runs in the CPU suite."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gaussianimage_amd", "csrc", "libgi2d.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    objdump, bundler = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "clang-offload-bundler")
    if not (os.path.exists(objdump) and os.path.exists(bundler) and shutil.which("objcopy")):
        pytest.skip("LLVM binary tools of the ROCm image not found")
    tmp = tmp_path_factory.mktemp("isa")
    fat = str(tmp / "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(MAGIC, blob)]
    assert starts, "no offload bundle in the library"
    text = []
    for k, a in enumerate(starts):
        part, co = str(tmp / f"part{k}.bin"), str(tmp / f"part{k}.co")
        open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.check_call([bundler, "--unbundle", "--type=o", f"--targets={TARGET}",
                               f"--input={part}", f"--output={co}"])
        text.append(subprocess.check_output([objdump, "-d", co], text=True))
    return text


def _mnemonics(text):
    for line in text.splitlines():
        m = re.match(r"\s+([a-z][a-z0-9_]+)\s", line)
        if m:
            yield m.group(1)


def test_one_code_object_per_translation_unit(disassembly):
    from gaussianimage_amd import build
    assert len(disassembly) == len(build.SOURCES)
    assert sum(t.count("v_exp_f32") for t in disassembly) > 10  # it is the rasterisers' code


def test_no_packed_fp32_no_scratch_no_scalar_writes(disassembly):
    bad = re.compile(r"^(v_pk_(add|mul|fma)_f32|scratch_.*|s_store_.*|s_buffer_store_.*|"
                     r"s_atomic_.*|s_buffer_atomic_.*|s_dcache_wb.*|s_scratch_store.*)$")
    found = sorted({m for t in disassembly for m in _mnemonics(t) if bad.match(m)})
    assert not found, found


def test_projection_has_no_double_precision_fma(disassembly):
    found = sorted({m for t in disassembly for m in _mnemonics(t)
                    if m.startswith("v_fma_f64") or m.startswith("v_fmac_f64")})
    assert not found, found
