# Backward-sensitive numbers of four workloads (no tests): for A/B runs of a build variant.
set -e
cd $GRAFT_REPO_ROOT
for WL in kodak5k kodak30k kodak30k_fitted div2k50k uhd200k; do
  timeout -k 10 200 python -u bench.py --steps 100 --warmup 10 --workload $WL --no-cpu-baseline --no-concurrent > gpurun_out/bwd_quick_$WL.json 2> gpurun_out/bwd_quick_$WL.err || { tail -20 gpurun_out/bwd_quick_$WL.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bwd_quick_$WL.json")); e=d["extras"]
print("$WL", "fps=%.0f" % d["value"], "fwd_bwd_us=%.1f" % (e["fwd_bwd_ms"]*1e3), "bwd_us=%.1f" % (e["bwd_only_ms"]*1e3), "fused_us=%.1f" % (e["fused_train_step_ms"]*1e3))
PY
done
