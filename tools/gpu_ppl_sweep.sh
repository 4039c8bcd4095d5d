# Forward-only timing of the rasteriser's workgroup shapes on several workloads.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ppl
for wl in kodak30k div2k50k uhd200k kodak30k_fitted; do
  for ppl in 1 2 4; do
    timeout -k 10 120 python bench.py --workload $wl --raster-ppl $ppl --steps 300 --no-extras --no-cpu-baseline > gpurun_out/ppl/${wl}_$ppl.json 2> gpurun_out/ppl/${wl}_$ppl.err
    python - <<PY
import json
d = json.load(open("gpurun_out/ppl/${wl}_$ppl.json"))
print("${wl} ppl=$ppl fps=%.0f ms=%.4f raster_ms=%.4f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]), flush=True)
PY
  done
done
