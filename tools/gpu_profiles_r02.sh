# Everything profiles/r02_* is made from, in one GPU call:
#  1. the driver's bench command -> gpurun_out/r02/bench_driver.json
#  2. rocprofv3 kernel trace + stats of the headline part of that command (no extras: the same
#     kernel names also run single frames in the extras) and of the full command
#  3. counters-only passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix) of the headline part
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 300 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_driver.json"))
r = d["roofline"]
print("driver cmd: value %.0f frames/s ms/step %.4f regions %d raster %.2f us frac %.3f" % (
    d["value"], d["ms_per_step"], d["timed_regions"], r["avg_launch_ms"] * 1e3, r["frac"]))
PY
cd /tmp && export TMPDIR=/tmp
HEAD="--gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o headline -- python3 $R/bench.py $HEAD > $O/headline.log 2>&1 || { tail -20 $O/headline.log; exit 1; }
python3 $R/tools/kstats.py $O/headline_kernel_stats.csv | head -6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o full -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/full.log 2>&1 || { tail -20 $O/full.log; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O -o pmc_$C -- python3 $R/bench.py $HEAD > $O/pmc_$C.log 2>&1 || { tail -20 $O/pmc_$C.log; exit 1; }
done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O -o pmc_SQ -- python3 $R/bench.py $HEAD > $O/pmc_SQ.log 2>&1 || { tail -20 $O/pmc_SQ.log; exit 1; }
cd $R
python3 tools/counters_json.py $O kodak30k_batch24 "tools/gpu_profiles_r02.sh: rocprofv3 --pmc passes of 'bench.py $HEAD'"
ls $O | head -40
