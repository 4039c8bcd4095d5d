# Kernel trace of the headline forward for several values of an environment variable the library
# was temporarily built to read (tuning experiments): gpu_try_env.sh VAR v1 v2 ...
set -e
R=$GRAFT_REPO_ROOT
VAR=$1; shift
mkdir -p $R/gpurun_out/tryenv
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tryenv -o t_$v -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/tryenv/t_$v.log 2>&1 || { tail -20 $R/gpurun_out/tryenv/t_$v.log; exit 1; }
  echo "== $VAR=$v"; python3 $R/tools/kstats.py $R/gpurun_out/tryenv/t_${v}_kernel_stats.csv | head -2 | cut -c1-110
done
done
