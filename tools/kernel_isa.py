"""Prints register / LDS / scratch use of the kernels in libgi2d.so whose (mangled) name contains
a given substring, and optionally their disassembly:  python tools/kernel_isa.py <substr> [-d]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gaussianimage_amd", "csrc", "libgi2d.so")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
want, dis = sys.argv[1], "-d" in sys.argv[2:]
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
blob = open(fat, "rb").read()
starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
for k, a in enumerate(starts):
    part, co = os.path.join(tmp, f"p{k}.bin"), os.path.join(tmp, f"p{k}.co")
    open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           f"--targets={TARGET}", f"--input={part}", f"--output={co}"])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    for blk in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or want not in name.group(1):
            continue
        f = {k_: re.search(r"\.%s:\s+(\S+)" % k_, blk) for k_ in
             ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
              "vgpr_spill_count")}
        print(name.group(1), {k_: (v.group(1) if v else None) for k_, v in f.items()})
        if dis:
            text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d",
                                            "--disassemble-symbols=" + name.group(1), co], text=True)
            print(text)
