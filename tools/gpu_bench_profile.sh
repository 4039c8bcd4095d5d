# Runs the default bench and a rocprofv3 kernel-trace of the same command on the GPU box.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
timeout -k 10 240 python -u bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o kt -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $R/gpurun_out/prof/run.log 2>&1 || { tail -20 $R/gpurun_out/prof/run.log; exit 1; }
ls -R $R/gpurun_out/prof | head -30
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o full -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-concurrent > $R/gpurun_out/prof/run_full.log 2>&1 || { tail -20 $R/gpurun_out/prof/run_full.log; exit 1; }
