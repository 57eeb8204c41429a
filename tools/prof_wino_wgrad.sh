# per-kernel times of the Winograd weight gradient (main kernel vs slab reduction) on a few bench shapes
set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r02
rm -rf $R/gpurun_out/r02/prof_ww
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/prof_ww -- python3 tools/ab_wino_wgrad.py 20 > $R/gpurun_out/r02/prof_ww.log 2>&1 || tail -5 $R/gpurun_out/r02/prof_ww.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02/prof_ww/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
