#!/bin/bash
# PMC passes over the matcher (run on the GPU box from the repo root):
#   tools/pmc_matcher.sh VARIANT [n] [desc_bytes] [splits]   -> gpurun_out/pmc_matcher_vVARIANT.txt
set -e
export TMPDIR=/tmp
V=$1; N=${2:-8192}; D=${3:-61}; S=${4:-0}
OUT=gpurun_out/pmc_m$V
rm -rf $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 tools/run_matcher_once.py $V $N $D $S 12 > /dev/null 2>&1 || echo "group $i failed"
done
python3 tools/pmc_summary.py $OUT/g1 $OUT/g2 $OUT/g3 $OUT/g4 $OUT/g5 | grep -i "hamming\|merge\|pack" > gpurun_out/pmc_matcher_v$V.txt
cat gpurun_out/pmc_matcher_v$V.txt
