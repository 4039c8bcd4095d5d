# Everything the committed profiles are made from, in one GPU call: the default bench line, the
# kernel traces (forward-only and full), the HBM-traffic passes and the instruction-mix passes.
# Afterwards, on the CPU side: tools/pmc_traffic.py, tools/pmc_sq_json.py <tag>, and copy the
# kernel_stats CSVs into profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu_bench_profile.sh > gpurun_out/refresh_bench.log 2>&1 || { tail -30 gpurun_out/refresh_bench.log; exit 1; }
echo "bench + traces done"
bash tools/gpu_pmc.sh > gpurun_out/refresh_pmc.log 2>&1 || { tail -30 gpurun_out/refresh_pmc.log; exit 1; }
echo "traffic passes done"
bash tools/gpu_pmc_sq.sh kodak30k bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/refresh_sq.log 2>&1 || { tail -30 gpurun_out/refresh_sq.log; exit 1; }
bash tools/gpu_pmc_sq.sh uhd200k bench.py --workload uhd200k --steps 30 --warmup 5 --no-cpu-baseline --no-extras >> gpurun_out/refresh_sq.log 2>&1 || { tail -30 gpurun_out/refresh_sq.log; exit 1; }
bash tools/gpu_pmc_sq.sh fit30k tools/fit_demo.py --iters 60 --points 30000 --height 512 --width 768 >> gpurun_out/refresh_sq.log 2>&1 || { tail -30 gpurun_out/refresh_sq.log; exit 1; }
echo "instruction-mix passes done"
head -c 600 gpurun_out/bench.json
