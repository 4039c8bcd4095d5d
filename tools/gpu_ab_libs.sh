# A/B of two builds of the library on ONE box: ab_libs/libA.so and ab_libs/libB.so take turns as
# csrc/libgi2d.so; kernel trace of a script each time.  gpu_ab_libs.sh <reps> <script.py> [args]
set -e
R=$GRAFT_REPO_ROOT
REPS=$1; shift
SCRIPT=$1; shift
export GI2D_ALLOW_STALE=1
mkdir -p $R/gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in A B; do
    cp $R/ab_libs/lib$v.so $R/gaussianimage_amd/csrc/libgi2d.so
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab -o ${v}_$rep -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/ab/${v}_$rep.log 2>&1 || { tail -20 $R/gpurun_out/ab/${v}_$rep.log; exit 1; }
    echo "== $v $rep"; python3 $R/tools/kstats.py $R/gpurun_out/ab/${v}_${rep}_kernel_stats.csv | head -${TOP:-2} | cut -c1-110
  done
done
