# Kernel trace of forward frames with spatially ordered Gaussians.  usage: gpu_sorted_kt.sh [workload]
set -e
R=$GRAFT_REPO_ROOT
WL=${1:-uhd200k}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o sorted_$WL -- python3 $R/tools/sorted_forward.py $WL > $R/gpurun_out/prof/sorted_$WL.log 2>&1 || { tail -20 $R/gpurun_out/prof/sorted_$WL.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/sorted_${WL}_kernel_stats.csv > $R/gpurun_out/prof/sorted_$WL.txt
sed -n 1,3p $R/gpurun_out/prof/sorted_$WL.txt
