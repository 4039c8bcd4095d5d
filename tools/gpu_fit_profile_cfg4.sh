# Kernel trace and instruction mix of the fused fitting step at ~1020x768 / 50k (initial regime).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fitc4 -- python3 $R/tools/fit_demo.py --iters 300 > $R/gpurun_out/prof/run_fitc4.log 2>&1 || { tail -20 $R/gpurun_out/prof/run_fitc4.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/fitc4_kernel_stats.csv > $R/gpurun_out/prof/fitc4.txt
sed -n 1,3p $R/gpurun_out/prof/fitc4.txt
cd $R && bash tools/gpu_pmc_sq.sh fitc4 tools/fit_demo.py --iters 60
