# GPU test suite, then the default bench without the CPU baseline; logs under gpurun_out/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 420 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -20 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_quick.json"))
print("fps", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "raster us", round(d["roofline"]["avg_launch_ms"] * 1e3, 2), "frac", round(d["roofline"]["frac"], 3))
for k, v in d.get("kernels", {}).items():
    print(" ", k, round(v["avg_launch_ms"] * 1e3, 2), "us frac", round(v["frac"], 3))
ts = d.get("train_step", {})
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in ts.items() if k in ("fused_ms", "fwd_bwd_ms", "reference_style_ms")}, ts.get("fused_batch"))
ex = d.get("extras", {})
print({k: round(ex[k]) for k in ("single_frame_fps", "reference_fps_loop", "decode_fps_fused") if k in ex})
PY
