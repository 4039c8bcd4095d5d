set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
tail -45 gpurun_out/pytest_gpu.log
timeout -k 10 120 python -u __graft_entry__.py smoke 2>&1 | tail -3
