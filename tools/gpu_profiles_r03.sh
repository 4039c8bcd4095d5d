# Everything profiles/r03_* is made from.  Two GPU calls (each within gpurun's limit):
#   gpu_profiles_r03.sh headline   -- the driver's bench command, its kernel traces, counters
#   gpu_profiles_r03.sh configs    -- every other workload: bench line, kernel trace of the
#                                     headline part, instruction-mix counters
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
HEAD="--gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
SQ_MIX="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
SQ_STALL="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
if [ "$1" = "headline" ]; then
  cd $R
  timeout -k 10 400 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_driver.json"))
r = d["roofline"]
print("driver cmd: value %.0f frames/s ms/step %.4f regions %d raster %.2f us frac %.3f cpu %.1f fps" % (
    d["value"], d["ms_per_step"], d["timed_regions"], r["avg_launch_ms"] * 1e3, r["frac"],
    d["cpu_baseline"]["value"]))
PY
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o headline -- python3 $R/bench.py $HEAD > $O/headline.log 2>&1 || { tail -20 $O/headline.log; exit 1; }
  python3 $R/tools/kstats.py $O/headline_kernel_stats.csv | head -4
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o full -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/full.log 2>&1 || { tail -20 $O/full.log; exit 1; }
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O -o pmc_$C -- python3 $R/bench.py $HEAD > $O/pmc_$C.log 2>&1 || { tail -20 $O/pmc_$C.log; exit 1; }
  done
  timeout -s KILL 120 rocprofv3 --pmc $SQ_MIX --output-format csv -d $O -o pmc_SQ -- python3 $R/bench.py $HEAD > $O/pmc_SQ.log 2>&1 || { tail -20 $O/pmc_SQ.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc $SQ_STALL --output-format csv -d $O -o pmc_SQstall -- python3 $R/bench.py $HEAD > $O/pmc_SQstall.log 2>&1 || { tail -20 $O/pmc_SQstall.log; exit 1; }
  cd $R
  python3 tools/counters_json.py $O kodak30k_batch24 "tools/gpu_profiles_r03.sh headline: rocprofv3 --pmc passes of 'bench.py $HEAD'" r03_counters.json
else
  mkdir -p $O/configs
  for WL in kodak5k kodak30k_rs div2k50k uhd200k kodak30k_fitted; do
    cd $R
    timeout -k 10 400 python -u bench.py --steps 50 --warmup 10 --workload $WL --no-cpu-baseline > $O/configs/$WL.json 2> $O/configs/$WL.err || { tail -20 $O/configs/$WL.err; exit 1; }
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/configs -o kt_$WL -- python3 $R/bench.py --workload $WL $HEAD > $O/configs/kt_$WL.log 2>&1 || { tail -20 $O/configs/kt_$WL.log; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc $SQ_MIX --output-format csv -d $O/configs -o pmc_$WL -- python3 $R/bench.py --workload $WL $HEAD > $O/configs/pmc_$WL.log 2>&1 || { tail -20 $O/configs/pmc_$WL.log; exit 1; }
    cd $R
    python3 - <<PY
import json
d = json.load(open("gpurun_out/r03/configs/$WL.json"))
r, t = d["roofline"], d.get("train_step", {})
print("$WL", "fps=%.0f" % d["value"], "raster_us=%.2f" % (r["avg_launch_ms"] * 1e3), "frac=%.3f" % r["frac"],
      "fused_ms=%.4f" % t.get("fused_ms", -1), "batch", (t.get("fused_batch") or {}).get("ms_per_image_iteration"))
PY
    python3 tools/kstats.py $O/configs/kt_${WL}_kernel_stats.csv | head -3
  done
fi
