#!/bin/bash
# Kernel-trace summary of the headline forward with one list-filling form: $1 (default 4);
# further arguments go to bench.py (e.g. --workload uhd200k).
set -e
cd $GRAFT_REPO_ROOT
FORM=${1:-4}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_form$FORM$(echo "$@" | tr -d " -")
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --project-form $FORM "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("form $FORM $@: value", round(d["value"]), "ms/step", d["ms_per_step"])
for r in csv.DictReader(open("$OUT/t_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3, 2), "us")
PY
