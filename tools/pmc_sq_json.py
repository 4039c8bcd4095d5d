"""Turns a counters-only pass of tools/gpu_pmc_sq.sh (``sq_<tag>_counter_collection.csv``) into
per-kernel averages and stores them under ``<tag>`` in profiles/r01_sq_instruction_mix.json
(bench.py reads the forward rasteriser's SQ_INSTS_VALU from there for roofline.valu_issue).
usage: pmc_sq_json.py <tag> [dir]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc_sq"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(root, "**", f"sq_{tag}_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        name = row["Kernel_Name"].split("(")[0]
        if "at::" in name or "rocclr" in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    out[k] = {c: round(sum(x) / len(x)) for c, x in sorted(d.items())}
    out[k]["launches"] = len(next(iter(d.values())))
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                   "r01_sq_instruction_mix.json")
data = json.load(open(dst)) if os.path.exists(dst) else {}
data[tag] = out
json.dump(data, open(dst, "w"), indent=1)
for k, v in out.items():
    print(f"{k[-56:]:56s} VALU {v.get('SQ_INSTS_VALU', 0):9d}  waves {v.get('SQ_WAVES', 0):6d}")
