# Evidence for one bf16 BASELINE configuration: bench line (eager + HIP-graph side object), kernel trace, per-shape conv table,
# MFMA-pipe busy of the conv kernels (separate --pmc pass).   tools/prof_cfg.sh <config 2|3|4> <tag>  ->  gpurun_out/<tag>/cfg<k>/
set -e
export TMPDIR=/tmp
C=${1:-3}
TAG=${2:-r03}
O=$PWD/gpurun_out/$TAG/cfg$C
mkdir -p $O
python3 bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || tail -3 $O/bench.err
echo "bench done"
rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-graph-side --conv-table $O/conv_shape_table.txt > $O/trace.log 2>&1 || tail -5 $O/trace.log
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/trace
echo "trace done"
rm -rf $O/pmc_MFMA
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_MFMA -- python3 bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph-side > $O/pmc_MFMA.log 2>&1 || tail -5 $O/pmc_MFMA.log
echo "pmc mfma done"
python3 tools/prof_cfg_post.py $O
rm -rf $O/pmc_MFMA
