# After a change to the backward: the GPU test suite, then the numbers that move -- fitted-size
# and model-init fused steps and forward+backward, and the config-4 fit.
set -e
cd $GRAFT_REPO_ROOT
bash tools/gpu_tests.sh > gpurun_out/bwd_check_tests.log 2>&1 || { tail -40 gpurun_out/bwd_check_tests.log; exit 1; }
tail -3 gpurun_out/bwd_check_tests.log
for WL in kodak30k kodak30k_fitted div2k50k uhd200k; do
  timeout -k 10 200 python -u bench.py --steps 100 --warmup 10 --workload $WL --no-cpu-baseline --no-concurrent > gpurun_out/bwd_check_$WL.json 2> gpurun_out/bwd_check_$WL.err || { tail -20 gpurun_out/bwd_check_$WL.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/bwd_check_$WL.json")); e=d["extras"]
print("$WL", "fps=%.0f" % d["value"], "fwd_bwd_us=%.1f" % (e["fwd_bwd_ms"]*1e3), "bwd_us=%.1f" % (e["bwd_only_ms"]*1e3), "fused_us=%.1f" % (e["fused_train_step_ms"]*1e3))
PY
done
timeout -k 10 200 python -u tools/fit_demo.py > gpurun_out/fit_demo.log 2>&1
tail -1 gpurun_out/fit_demo.log | cut -c1-140
