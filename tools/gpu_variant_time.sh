# Kernel trace of the headline forward with the library at csrc/libgi2d.so (quick A/B of builds)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o ab_$1 -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "${@:2}" > $R/gpurun_out/prof/ab_$1.log 2>&1 || { tail -20 $R/gpurun_out/prof/ab_$1.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/ab_$1_kernel_stats.csv | head -3
