# Copies what tools/gpu_profiles_r03.sh left under gpurun_out/r03 into profiles/ (run here, after
# the GPU calls): collect_r03_profiles.sh headline | configs
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03
if [ "$1" = "headline" ]; then
  cp $O/bench_driver.json profiles/r03_bench_driver_cmd.json
  cp $O/headline_kernel_stats.csv profiles/r03_headline_kernel_stats.csv
  cp $O/full_kernel_stats.csv profiles/r03_full_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE SQ SQstall; do cp $O/pmc_${c}_counter_collection.csv profiles/r03_pmc_$c.csv; done
  T=$(mktemp -d); cp $O/pmc_*_counter_collection.csv $T/
  python3 tools/counters_json.py $T kodak30k_batch24 "tools/gpu_profiles_r03.sh headline: rocprofv3 --pmc passes of 'bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline'" r03_counters.json > /dev/null
  rm -rf $T
else
  for WL in kodak5k kodak30k_rs div2k50k uhd200k kodak30k_fitted; do
    cp $O/configs/$WL.json profiles/r03_configs/$WL.json
    cp $O/configs/kt_${WL}_kernel_stats.csv profiles/r03_configs/${WL}_headline_kernel_stats.csv
    cp $O/configs/pmc_${WL}_counter_collection.csv profiles/r03_configs/${WL}_pmc_SQ.csv
    T=$(mktemp -d); cp $O/configs/pmc_${WL}_counter_collection.csv $T/
    B=24; [ $WL = uhd200k ] && B=1
    python3 tools/counters_json.py $T ${WL}_batch$B "tools/gpu_profiles_r03.sh configs: rocprofv3 --pmc pass of 'bench.py --workload $WL --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline'" r03_counters.json > /dev/null
    rm -rf $T
  done
fi
