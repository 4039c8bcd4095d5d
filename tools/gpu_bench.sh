# The default bench line (what the driver runs) -> gpurun_out/bench.json, then its kernel trace.
# usage: gpu_bench.sh [bench args...]
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof
timeout -k 10 400 python -u bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
r = d["roofline"]
print("value %.0f frames/s  ms/step %.4f  regions %d  raster %.2f us  frac %.3f (held %.3f)" % (
    d["value"], d["ms_per_step"], d["timed_regions"], r["avg_launch_ms"] * 1e3, r["frac"], r["frac_lists_held"]))
for k, v in d.get("kernels", {}).items():
    print("  %-34s %8.2f us  %7.0f GB/s  frac %.3f" % (k, v["avg_launch_ms"] * 1e3, v["achieved"], v["frac"]))
t = d.get("train_step")
if t:
    print("  train_step: fused %.4f ms  reference-style %.4f ms  fwd+bwd %.4f ms  frac %.3f" % (
        t["fused_ms"], t["reference_style_ms"], t["fwd_bwd_ms"], t["roofline"]["frac"]))
for k, v in d.get("extras", {}).items():
    print("  ", k, v if not isinstance(v, float) else round(v, 4))
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
