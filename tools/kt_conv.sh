# kernel trace of one conv layer (tools/run_conv.py args): per-kernel average durations
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02/kt_conv
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/run_conv.py "$@" > $O.log 2>&1 || tail -5 $O.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02/kt_conv/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
