"""Counter passes of tools/gpu_profiles_r0N.sh -> profiles/<file> (default r02_counters.json; per-launch averages
of the batch kernels of the headline command; bench.py quotes them with their source).

HBM bytes as the MI355X guide prescribes: the counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read and is doubled (an upper estimate for narrow
gathers, whose width is not calibrated); WRITE_SIZE is exact for 16-byte streaming stores."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, key, source = sys.argv[1], sys.argv[2], sys.argv[3]
dst_name = sys.argv[4] if len(sys.argv) > 4 else "r02_counters.json"
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "pmc_*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))


def avg(kernel_part, counter, skip_first=2):
    vals = []
    for k, d in acc.items():
        if re.search(kernel_part, k) and counter in d:
            vals += d[counter]
    vals = vals[skip_first:] if len(vals) > skip_first + 4 else vals
    return sum(vals) / len(vals) if vals else None


out = {"source": source}
# (the batch launch's rasteriser: first pass, one colour set, records; any workgroup shape)
for tag, part in (("raster_fwd", r"rasterize_sum_fwd_kernel<true, false, 1, \d, true>"),
                  ("project", r"project_agg_fwd_kernel<false, false>"),
                  ("project_direct", r"project_cholesky_fwd_kernel<false>")):
    f, w = avg(part, "FETCH_SIZE"), avg(part, "WRITE_SIZE")
    if f is not None and w is not None:
        out[tag + "_fetch_bytes_x2"] = f * 1024 * 2
        out[tag + "_fetch_bytes_raw"] = f * 1024
        out[tag + "_write_bytes"] = w * 1024
        out[tag + "_hbm_bytes"] = f * 1024 * 2 + w * 1024
    for c, name in (("SQ_INSTS_VALU", "valu_instr"), ("SQ_INSTS_SALU", "salu_instr"),
                    ("SQ_INSTS_LDS", "lds_instr"), ("SQ_WAVES", "waves"),
                    ("SQ_WAVE_CYCLES", "wave_quad_cycles"), ("SQ_WAIT_ANY", "wait_any_quad_cycles"),
                    ("SQ_WAIT_INST_ANY", "wait_inst_any_quad_cycles"),
                    ("SQ_ACTIVE_INST_ANY", "active_inst_any_quad_cycles"),
                    ("SQ_INST_CYCLES_SALU", "salu_quad_cycles"),
                    ("SQ_ACTIVE_INST_LDS", "active_inst_lds_quad_cycles"),
                    ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles")):
        v = avg(part, c)
        if v is not None:
            out[tag + "_" + name] = v
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                   dst_name)
data = json.load(open(dst)) if os.path.exists(dst) else {}
data[key] = out
json.dump(data, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
