# kernel trace + stats of the default bench line: tools/prof_bench.sh <tag> [bench args]
set -e
export TMPDIR=/tmp
R=$PWD
TAG=$1; shift
mkdir -p $R/gpurun_out/r02
rm -rf $R/gpurun_out/r02/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/r02/prof_$TAG.log 2>&1 || tail -5 $R/gpurun_out/r02/prof_$TAG.log
cp $R/gpurun_out/r02/prof_$TAG/*/*kernel_stats.csv $R/gpurun_out/r02/kernel_stats_$TAG.csv
grep '^{' $R/gpurun_out/r02/prof_$TAG.log > $R/gpurun_out/r02/bench_prof_$TAG.json || true
rm -rf $R/gpurun_out/r02/prof_$TAG
