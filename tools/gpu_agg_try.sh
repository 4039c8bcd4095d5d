#!/bin/bash
# Aggregated projection: workgroups per batch (GI2D_AGG_WG, experiment knob) by kernel trace.
set -e
cd $GRAFT_REPO_ROOT
for WG in 128 168 192 256; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/agg_wg$WG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GI2D_AGG_WG=$WG timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, json
for r in csv.DictReader(open("$OUT/t_kernel_stats.csv")):
    if "project_agg" in r["Name"]:
        print("wg $WG", r["Calls"], round(float(r["AverageNs"])/1e3, 2), "us")
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("   value", round(d["value"]), d["ms_per_step"])
PY
done
