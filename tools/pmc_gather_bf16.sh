# What holds the bf16 gather kernel back: counter passes (MFMA / LDS / issue mix) for one shape through tools/conv_microbench.py.
#   tools/pmc_gather_bf16.sh <N Cin H W Cout k stride pad mode iters which>      -> gpurun_out/pgb_summary.txt
set -e
export TMPDIR=/tmp
export C2M_BENCH_BF16=1
R=$PWD
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  # (a TA_* pass -- TA_TA_BUSY_sum, TA_ADDR_STALLED_BY_TC_CYCLES_sum ... -- made rocprofv3 abort with signal 6 and hang on this pool: SQ only)
  n=$(echo $C | cut -c1-14 | tr ' ' '_')
  rm -rf $R/gpurun_out/pgb_$n
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pgb_$n -- python3 tools/conv_microbench.py "$@" > gpurun_out/pgb_$n.log 2>&1 || tail -3 gpurun_out/pgb_$n.log
done
python3 - <<'PY' > gpurun_out/pgb_summary.txt
import csv, glob, collections
per = collections.defaultdict(dict)
for f in glob.glob('gpurun_out/pgb_*/*/*_counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_' in k:
            agg[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        for c, v in cs.items():
            per[k][c] = sum(v) / len(v)
for k, m in per.items():
    gui = m.get('GRBM_GUI_ACTIVE', 0) / 8 or 1
    print("==", k, " kernel cycles/XCD", round(gui))
    for c in sorted(m):
        print(f"  {c:40s} {m[c]:16.0f}  per-SIMD-cycle {m[c] / 1024 / gui:9.4f}  per-CU-cycle {m[c] / 256 / gui:9.4f}")
PY
cat gpurun_out/pgb_summary.txt
