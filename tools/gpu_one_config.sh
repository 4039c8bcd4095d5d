# The bench line of one workload -> gpurun_out/configs/<workload>.json
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/configs
WL=${1:-kodak30k}
timeout -k 10 300 python -u bench.py --steps 50 --warmup 10 --workload $WL --no-cpu-baseline > gpurun_out/configs/$WL.json 2> gpurun_out/configs/$WL.err || { tail -20 gpurun_out/configs/$WL.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/configs/$WL.json"))
t = d["train_step"]
print("$WL", round(d["value"]), "fused_ms", t["fused_ms"], "fused_batch", t.get("fused_batch"))
PY
