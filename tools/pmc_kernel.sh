# Counter passes for one conv shape: tools/pmc_kernel.sh <kernel substring> <conv_microbench args>
set -e
export TMPDIR=/tmp
R=$PWD
K=$1; shift
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH"; do
  n=$(echo $C | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/pk_$n
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pk_$n -- python3 tools/conv_microbench.py "$@" > gpurun_out/pk_$n.log 2>&1 || tail -3 gpurun_out/pk_$n.log
done
python3 - "$K" <<'PY'
import csv, glob, collections, sys
K=sys.argv[1]
m={}
for f in glob.glob('gpurun_out/pk_*/*/*_counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if K in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in agg.items(): m[c]=sum(v)/len(v)
gui=m['GRBM_GUI_ACTIVE']/8
print("kernel cycles/XCD", round(gui), "MFMA busy", round(m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/gui,3))
for c in sorted(m): print(f"  {c:28s} {m[c]:16.0f}  per-SIMD-cycle {m[c]/1024/gui:8.3f}")
PY
