# rocprofv3 kernel stats of bench.py for one workload: bash tools/gpu_profile_wl.sh <workload>
set -e
R=$GRAFT_REPO_ROOT
WL=${1:-uhd200k}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o wl_$WL -- python3 $R/bench.py --steps 50 --warmup 5 --workload $WL --no-cpu-baseline $EXTRA > $R/gpurun_out/prof/run_$WL.log 2>&1 || { tail -20 $R/gpurun_out/prof/run_$WL.log; exit 1; }
