# Kernel trace of the headline forward for several sets of bench.py arguments on one box:
# gpu_bench_args.sh "args1" "args2" ...
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/bargs
cd /tmp && export TMPDIR=/tmp
i=0
for rep in 1 2; do
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bargs -o t_$i -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $a > $R/gpurun_out/bargs/t_$i.log 2>&1 || { tail -20 $R/gpurun_out/bargs/t_$i.log; exit 1; }
  echo "== $a"; python3 $R/tools/kstats.py $R/gpurun_out/bargs/t_${i}_kernel_stats.csv | head -2 | cut -c1-110; grep '^{"metric"' $R/gpurun_out/bargs/t_$i.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   fps %.0f ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
done
