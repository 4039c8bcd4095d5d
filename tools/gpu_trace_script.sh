# Kernel trace of a python script of this repo: gpu_trace_script.sh <tag> <script.py> [args...]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$1; shift
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o $TAG -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/prof/$TAG.log 2>&1 || { tail -20 $R/gpurun_out/prof/$TAG.log; exit 1; }
tail -2 $R/gpurun_out/prof/$TAG.log
python3 $R/tools/kstats.py $R/gpurun_out/prof/${TAG}_kernel_stats.csv | head -${TOP:-8}
