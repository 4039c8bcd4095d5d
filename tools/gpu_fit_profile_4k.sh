# Kernel trace and instruction mix of the fused fitting step at 3840x2160 / 200k.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fit4k -- python3 $R/tools/fit_demo.py --iters 200 --points 200000 --height 2160 --width 3840 > $R/gpurun_out/prof/run_fit4k.log 2>&1 || { tail -20 $R/gpurun_out/prof/run_fit4k.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/fit4k_kernel_stats.csv > $R/gpurun_out/prof/fit4k.txt
sed -n 1,4p $R/gpurun_out/prof/fit4k.txt
cd $R && bash tools/gpu_pmc_sq.sh fit4k tools/fit_demo.py --iters 40 --points 200000 --height 2160 --width 3840
