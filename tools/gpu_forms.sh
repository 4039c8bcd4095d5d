# Bench line of one workload for every list-filling form of the projection (GI2D_OPT_PROJECT_AGG):
# gpu_forms.sh <workload> [forms...]
set -e
cd $GRAFT_REPO_ROOT
WL=$1; shift
mkdir -p gpurun_out/forms
for F in "$@"; do
  timeout -k 10 200 python bench.py --workload $WL --steps 100 --warmup 10 --no-cpu-baseline --no-extras --project-form $F > gpurun_out/forms/${WL}_$F.json 2> gpurun_out/forms/${WL}_$F.err || { tail -5 gpurun_out/forms/${WL}_$F.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/forms/${WL}_$F.json"))
k = d.get("kernels", {})
print("$WL form $F: fps %.0f ms/step %.4f raster %.2f us" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"] * 1e3),
      {n: round(v["avg_launch_ms"] * 1e3, 2) for n, v in k.items() if "project" in n})
PY
done
