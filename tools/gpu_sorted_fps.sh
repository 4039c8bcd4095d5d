# Forward FPS with the Gaussians in the model's order and in tile order, three workloads.
set -e
cd $GRAFT_REPO_ROOT
for WL in kodak5k kodak30k div2k50k uhd200k; do
  timeout -k 10 200 python -u bench.py --steps 300 --warmup 20 --workload $WL --no-cpu-baseline --no-concurrent > gpurun_out/sorted_$WL.json 2> gpurun_out/sorted_$WL.err || { tail -20 gpurun_out/sorted_$WL.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sorted_$WL.json")); e=d["extras"]
print("$WL", "fps=%.0f" % e["forward_fps_no_event_brackets"], "tile order fps=%.0f" % e["forward_fps_spatial_order"], "fused_us=%.1f / %.1f" % (e["fused_train_step_ms"]*1e3, e["fused_train_step_ms_spatial_order"]*1e3))
PY
done
