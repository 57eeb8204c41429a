# PMC passes over tools/ab_wino_wgrad.py (first AB_SHAPES shapes) or over PMC_CMD="script args": clock, MFMA-pipe busy, wait / issue-stall split, instruction mix
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/${PMC_OUT:-gpurun_out/r02/pmc_ab}
rm -rf $O; mkdir -p $O
export AB_SHAPES=${AB_SHAPES:-2}
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 ${PMC_CMD:-tools/ab_wino_wgrad.py 10} > $O/p1.log 2>&1 || tail -5 $O/p1.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -- python3 ${PMC_CMD:-tools/ab_wino_wgrad.py 10} > $O/p2.log 2>&1 || tail -5 $O/p2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 ${PMC_CMD:-tools/ab_wino_wgrad.py 10} > $O/kt.log 2>&1 || tail -5 $O/kt.log
python3 - <<'PY'
import csv, glob, collections
import os
O = os.environ.get("PMC_OUT", "gpurun_out/r02/pmc_ab")
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{O}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"): n[k] += 1
    return acc, n
a1, n1 = load("p1"); a2, n2 = load("p2")
dur = {}
for f in glob.glob(f"{O}/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)): dur[r["Name"].split("(")[0][:60]] = float(r["AverageNs"])
for k in a1:
    if "mfma" not in k and "conv" not in k and "wino" not in k: continue
    c = a1[k]; d = a2.get(k, {})
    if c["SQ_INSTS_MFMA"] == 0: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8 / n1[k]
    ghz = cyc / dur[k] if k in dur else float("nan")
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    wc = c["SQ_WAVE_CYCLES"]
    mf = c["SQ_INSTS_MFMA"]
    print(f"{k}: launches {n1[k]} avg {dur.get(k, 0)/1e3:.1f} us  clock(profiled) {ghz:.2f} GHz  MFMA busy {busy:.3f}  "
          f"wait_any {c['SQ_WAIT_ANY']/wc:.3f} issue_stall {c['SQ_WAIT_INST_ANY']/wc:.3f} active {c['SQ_ACTIVE_INST_ANY']/wc:.3f} | per MFMA: "
          f"VALU {d.get('SQ_INSTS_VALU',0)/mf:.2f} SALU {d.get('SQ_INSTS_SALU',0)/mf:.2f} LDS {d.get('SQ_INSTS_LDS',0)/mf:.2f} VMEMrd {d.get('SQ_INSTS_VMEM_RD',0)/mf:.3f} "
          f"| wait_inst_lds/wave_cyc {d.get('SQ_WAIT_INST_LDS',0)/wc:.3f} active_valu {d.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f} active_lds {d.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f} lds_bank_conflict/wave_cyc {d.get('SQ_LDS_BANK_CONFLICT',0)/wc:.3f}")
PY
