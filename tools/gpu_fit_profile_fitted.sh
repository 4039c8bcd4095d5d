# Kernel trace and instruction mix of the fused fitting step with fitted-size supports
# (config 2 shape, initial Cholesky factors x3).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fitx3 -- python3 $R/tools/fit_demo.py --iters 300 --points 30000 --height 512 --width 768 --footprint 3 > $R/gpurun_out/prof/run_fitx3.log 2>&1 || { tail -20 $R/gpurun_out/prof/run_fitx3.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/fitx3_kernel_stats.csv > $R/gpurun_out/prof/fitx3.txt
sed -n 1,4p $R/gpurun_out/prof/fitx3.txt
cd $R && bash tools/gpu_pmc_sq.sh fitx3 tools/fit_demo.py --iters 60 --points 30000 --height 512 --width 768 --footprint 3
