# Quick per-kernel times of the forward path: rocprofv3 kernel trace of a short forward-only bench.
# usage: gpu_kt.sh [workload] [extra bench args...]
set -e
R=$GRAFT_REPO_ROOT
WL=${1:-kodak30k}; shift || true
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o kt_$WL -- python3 $R/bench.py --workload $WL --steps 300 --warmup 20 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/prof/kt_$WL.log 2>&1 || { tail -20 $R/gpurun_out/prof/kt_$WL.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/kt_${WL}_kernel_stats.csv > $R/gpurun_out/prof/kt_$WL.txt
sed -n 1,4p $R/gpurun_out/prof/kt_$WL.txt
grep -o '"value": [0-9.]*' $R/gpurun_out/prof/kt_$WL.log | sed -n 1p
