# Kernel trace of tools/batch_fit_try.py (batched fused fit at several batch sizes).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bfit -- python3 $R/tools/batch_fit_try.py "$@" > $R/gpurun_out/prof/bfit.log 2>&1 || { tail -20 $R/gpurun_out/prof/bfit.log; exit 1; }
python3 $R/tools/kstats.py $R/gpurun_out/prof/bfit_kernel_stats.csv > $R/gpurun_out/prof/bfit.txt
sed -n 1,12p $R/gpurun_out/prof/bfit.txt
