# Evidence for profiles/ (any round): bench line, kernel trace, HBM traffic (separate --pmc passes, calibrated), MFMA busy.
#   tools/prof_round.sh r03   ->  gpurun_out/r03/final/*   (copy the summaries into profiles/r03_*)
set -e
export TMPDIR=/tmp
TAG=${1:-r03}
R=$PWD
O=$R/gpurun_out/$TAG/final
mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err || tail -3 $O/bench_final.err
echo "bench done"
rm -rf $O/trace
# kernel durations of ISOLATED launches: the branch streams (ops.aux_branch / ops.deferred_wgrads) are off in event steps anyway; the
# two warm-up steps would run them (a launch sharing the chip with another stream's is stretched 2-10x) -> switched off for this run
C2M_AUX_STREAM=0 C2M_DEFER_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 2 --event-every 1 --no-cpu-baseline --side-configs '' --conv-table $O/conv_shape_table.txt > $O/trace.log 2>&1 || tail -5 $O/trace.log
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv
cp $O/trace/*/*kernel_trace.csv $O/kernel_trace.csv
rm -rf $O/trace
echo "trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C $O/cal_$C
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --side-configs '' > $O/pmc_$C.log 2>&1 || tail -5 $O/pmc_$C.log
  rocprofv3 --pmc $C --output-format csv -d $O/cal_$C -- python3 tools/pmc_calib.py > $O/cal_$C.log 2>&1 || tail -5 $O/cal_$C.log
  echo "pmc $C done"
done
rm -rf $O/pmc_MFMA
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_MFMA -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --side-configs '' > $O/pmc_MFMA.log 2>&1 || tail -5 $O/pmc_MFMA.log
echo "pmc mfma done"
python3 tools/prof_round2_post.py $O
python3 tools/timeline_occupancy.py $O/kernel_trace.csv 5 256 > $O/launches_per_step.txt 2>&1 || tail -2 $O/launches_per_step.txt   # exact launches per step (no one-off blits)
rm -f $O/kernel_trace.csv
