# Where a kernel's wave cycles go: one counters-only rocprofv3 pass (8 SQ counters) of a python
# program of this repo.  usage: gpu_pmc_stall.sh <tag> <script.py> [args...]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$1; shift
mkdir -p $R/gpurun_out/pmc_stall
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_stall -o st_$TAG -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/pmc_stall/st_$TAG.log 2>&1 || { tail -20 $R/gpurun_out/pmc_stall/st_$TAG.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$R/gpurun_out/pmc_stall/**/st_${TAG}_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "at::" in k or "rocclr" in k:
        continue
    v = {c: sum(x) / len(x) for c, x in d.items()}
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print(k, "launches", len(next(iter(d.values()))))
    print("    " + "  ".join("%s %.3g (%.0f%%)" % (c.replace("SQ_", ""), x, 100 * x / wc) for c, x in sorted(v.items())))
PY
