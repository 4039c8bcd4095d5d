# GPU test suite + the not-yet-collected test file given as $1 (optional) + smoke; logs under gpurun_out/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/pytest_gpu.log; exit 1; }
tail -15 gpurun_out/pytest_gpu.log
if [ -n "$1" ]; then
  timeout -k 10 300 python -u -m pytest "$1" -x -v --timeout 120 --timeout-method thread > gpurun_out/pytest_extra.log 2>&1 || { tail -60 gpurun_out/pytest_extra.log; exit 1; }
  tail -8 gpurun_out/pytest_extra.log
fi
timeout -k 10 120 python -u __graft_entry__.py smoke 2>&1 | tail -3
