# After a kernel change: GPU tests, then per-kernel times and VALU counts of the forward path at
# config 2 and at 3840x2160 / 200k.   usage: gpu_check.sh [notests]
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
if [ "$1" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/pytest_gpu.log; exit 1; }
  tail -3 gpurun_out/pytest_gpu.log
fi
for WL in kodak30k uhd200k; do
  bash tools/gpu_kt.sh $WL
  bash tools/gpu_pmc_sq.sh fwd_$WL bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-extras | grep -A1 "rasterize_sum_fwd\|project_" 
done
