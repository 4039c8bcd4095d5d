# HBM traffic counters of the bench's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# passes (they do not fit one pass on gfx950), counters only (no tracing domains).
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc -o $C -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc/$C.log 2>&1 || { tail -20 $R/gpurun_out/pmc/$C.log; exit 1; }
done
ls $R/gpurun_out/pmc
