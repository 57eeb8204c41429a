# MFMA-pipe utilisation of one conv shape: tools/pmc_mfma.sh <conv_microbench args>
set -e
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/pmc_mfma
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 tools/conv_microbench.py "$@" > gpurun_out/pmc_mfma.log 2>&1 || tail -5 gpurun_out/pmc_mfma.log
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_mfma/*/*_counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'conv_' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        m={c:sum(x)/len(x) for c,x in v.items()}
        gui=m.get('GRBM_GUI_ACTIVE',0)/8
        print(k, {c:round(x) for c,x in m.items()})
        if gui: print("   kernel cycles/XCD", round(gui), " MFMA busy frac", round(m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/gui,3))
PY
