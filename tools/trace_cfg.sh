# kernel trace of one BASELINE configuration: tools/trace_cfg.sh <config> <outdir>
set -e
export TMPDIR=/tmp
C=${1:-3}
O=$PWD/${2:-gpurun_out/r03/cfg$C}
mkdir -p $O
rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --conv-table $O/conv_shape_table.txt > $O/trace.log 2>&1 || tail -5 $O/trace.log
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/trace
tail -c 1500 $O/trace.log
