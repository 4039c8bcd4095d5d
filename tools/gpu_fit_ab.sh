# Kernel traces of the fused fitting step with the library as built (A/B of builds): config 2,
# fitted-size supports, config 4, and the batched fit; prints the fit kernels' average durations.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fit_ab_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $tag -- python3 "$@" > $O/$tag.log 2>&1 || { tail -20 $O/$tag.log; exit 1; }
  echo "== $tag"; python3 $R/tools/kstats.py $O/${tag}_kernel_stats.csv | grep -E "fit_raster|fit_tail|rasterize_sum_bwd|gather" | cut -c1-150
}
run cfg2 $R/tools/fit_demo.py --iters 300 --points 30000 --height 512 --width 768
run cfg2x3 $R/tools/fit_demo.py --iters 300 --points 30000 --height 512 --width 768 --footprint 3
run cfg4 $R/tools/fit_demo.py --iters 300
run bfit $R/tools/batch_fit_try.py
