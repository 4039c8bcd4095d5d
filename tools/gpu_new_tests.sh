# Runs the GPU tests named on the command line (pytest node ids or -k expressions); log under gpurun_out/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest "$@" -m gpu -x -v -s --timeout 420 --timeout-method thread > gpurun_out/pytest_new.log 2>&1 || { tail -80 gpurun_out/pytest_new.log; exit 1; }
grep -E "PASSED|FAILED|SKIPPED|config 4 full fit|passed|failed" gpurun_out/pytest_new.log | tail -40
