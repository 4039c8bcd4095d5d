# The bench line of every BASELINE configuration plus the fitted-like variant -> gpurun_out/configs/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/configs
for WL in kodak5k kodak30k kodak30k_rs div2k50k uhd200k kodak30k_fitted; do
  timeout -k 10 300 python -u bench.py --steps 50 --warmup 10 --workload $WL --no-cpu-baseline > gpurun_out/configs/$WL.json 2> gpurun_out/configs/$WL.err || { tail -20 gpurun_out/configs/$WL.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/configs/$WL.json"))
e=d.get("extras",{}); t=d.get("train_step",{}); r=d["roofline"]
print("$WL", "B=%d" % d["config"]["frames_per_step"], "fps=%.0f" % d["value"], "1frame_fps=%.0f" % e.get("single_frame_fps",-1),
      "raster_us=%.2f" % (r["avg_launch_ms"]*1e3), "frac=%.3f" % r["frac"], "I/frame=%d" % d["config"]["intersections_per_frame"],
      "fwd_bwd_ms=%.4f" % t.get("fwd_bwd_ms",-1), "ref_train_ms=%.3f" % t.get("reference_style_ms",-1),
      "fused_ms=%.4f" % t.get("fused_ms",-1), "ref_fps=%.0f" % e.get("reference_fps_loop",-1), "decode=%.0f" % e.get("decode_fps_fused",-1))
PY
done
