# Kernel-trace and instruction-mix of the fused fitting step at config 2 (clean: one stream).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fit -- python3 $R/tools/fit_demo.py --iters 300 --points 30000 --height 512 --width 768 > $R/gpurun_out/prof/run_fit.log 2>&1 || { tail -20 $R/gpurun_out/prof/run_fit.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/fit_kernel_stats.csv | head -8
cd $R && bash tools/gpu_pmc_sq.sh fit30k tools/fit_demo.py --iters 60 --points 30000 --height 512 --width 768
