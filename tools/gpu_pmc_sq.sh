# Instruction-mix counters (one rocprofv3 pass, counters only) of a python program of this repo.
# usage: gpu_pmc_sq.sh <tag> <script.py> [args...]     (script path relative to the repo root)
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$1; shift
mkdir -p $R/gpurun_out/pmc_sq
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -o sq_$TAG -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/pmc_sq/sq_$TAG.log 2>&1 || { tail -20 $R/gpurun_out/pmc_sq/sq_$TAG.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$R/gpurun_out/pmc_sq/**/sq_${TAG}_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "at::" in k or "rocclr" in k:
        continue
    v = {c: round(sum(x) / len(x)) for c, x in d.items()}
    print(k, "launches", len(next(iter(d.values()))))
    print("    waves %d  VALU %d (%.0f/wave, %.2f us at 614 G/s)  SALU %d  LDS %d  VMEM rd %d wr %d" % (
        v["SQ_WAVES"], v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1),
        v["SQ_INSTS_VALU"] / 614.4e3, v["SQ_INSTS_SALU"], v["SQ_INSTS_LDS"],
        v["SQ_INSTS_VMEM_RD"], v["SQ_INSTS_VMEM_WR"]))
PY
