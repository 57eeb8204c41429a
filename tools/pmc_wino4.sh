# SQ counter passes over tools/bench_wino4.py (the F(4x4,3x3) kernel next to the F(2x2,3x3) one)
set -e
export TMPDIR=/tmp
export W4_SHAPES=${W4_SHAPES:-"40,128,64,128,128,0;40,256,32,64,256,0"}     # (the program after -- must be python itself: no env wrapper)
R=$PWD
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  n=$(echo $C | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/pw4_$n
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pw4_$n -- python3 tools/bench_wino4.py 3 > gpurun_out/pw4_$n.log 2>&1 || tail -3 gpurun_out/pw4_$n.log
done
python3 - <<'PY' > gpurun_out/pw4_summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pw4_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_wino' in k:
            per[(k[:40], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, gs), cs in sorted(per.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    gui = m.get('GRBM_GUI_ACTIVE', 0) / 8 or 1
    w = m.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"== {k} grid {gs}: cycles {gui:.0f}  MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / gui:.3f}  "
          f"LDS conflict/active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}  "
          f"wait_inst_any/wave {m.get('SQ_WAIT_INST_ANY', 0) / w:.3f}  wait_lds/wave {m.get('SQ_WAIT_INST_LDS', 0) / w:.3f}  "
          f"valu active/wave {m.get('SQ_ACTIVE_INST_VALU', 0) / w:.3f}  lds active/wave {m.get('SQ_ACTIVE_INST_LDS', 0) / w:.3f}  "
          f"VALU/MFMA {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f}  LDS/MFMA {m.get('SQ_INSTS_LDS', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f}")
PY
cat gpurun_out/pw4_summary.txt
