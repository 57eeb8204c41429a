set -e
export TMPDIR=/tmp
R=$PWD
for m in 0 1 3; do
  O=$R/gpurun_out/r05/normprof_$m; rm -rf $O; mkdir -p $O
  C2M_NORM_FUSED=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --side-configs '' > $O/log 2>&1 || tail -3 $O/log
  cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/trace
  echo "== C2M_NORM_FUSED=$m"; python3 tools/kernel_groups.py $O/kernel_stats.csv 7 | grep -i "norm\|total"
  grep "ms_per_step" $O/log | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" || true
done
