#!/bin/bash
# The headline forward at several batch sizes (frames per step).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/batches
for B in "$@"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --batch $B > gpurun_out/batches/b$B.json 2> gpurun_out/batches/b$B.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/batches/b$B.json").read().strip().splitlines()[-1])
print("B=$B value", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "raster us", round(d["roofline"]["avg_launch_ms"]*1e3,1))
PY
done
