# Forward FPS / roofline of every BASELINE configuration plus the fitted-like variant.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/configs
for WL in kodak5k kodak30k kodak30k_rs div2k50k uhd200k kodak30k_fitted; do
  timeout -k 10 200 python -u bench.py --steps 100 --warmup 10 --workload $WL --no-cpu-baseline > gpurun_out/configs/$WL.json 2> gpurun_out/configs/$WL.err || { tail -20 gpurun_out/configs/$WL.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/configs/$WL.json"))
e=d.get("extras",{})
print("$WL", "fps=%.0f" % d["value"], "raster_ms=%.4f" % d["roofline"]["avg_launch_ms"], "frac=%.3f" % d["roofline"]["frac"], "I=%d" % d["config"]["intersections"], "fwd_bwd_ms=%.4f" % e.get("fwd_bwd_ms",-1), "train_ms=%.3f" % e.get("train_step_ms",-1), "fused_ms=%.4f" % e.get("fused_train_step_ms",-1), "ref_fps=%.0f" % e.get("reference_fps_loop",-1))
PY
done
