set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python tools/conv_microbench.py 40 256 32 64 256 3 1 1 zeros 10 fwd
python tools/conv_microbench.py 40 512 16 32 512 3 1 1 zeros 10 fwd
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS SQ_INST_LEVEL_LDS"; do
  n=$(echo $C | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$n -- python3 tools/conv_microbench.py 40 256 32 64 256 3 1 1 zeros 3 fwd > gpurun_out/pmc_$n.log 2>&1 || tail -5 gpurun_out/pmc_$n.log
done
ls gpurun_out/pmc_*/*/ | head
