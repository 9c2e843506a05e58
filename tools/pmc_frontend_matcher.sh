#!/bin/bash
# Counters of the front-end's two matcher launches (500 frames x 1500 features), two rocprofv3 --pmc passes,
# mean per dispatch:  bash tools/pmc_frontend_matcher.sh TAG  ->  gpurun_out/pmc_frontend500_matcher_TAG.txt
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/pmc_fm1 $OUT/pmc_fm2
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/pmc_fm1 -- python3 tools/profile_frontend.py 500 1500 > /dev/null 2> $OUT/pmc_fm.err || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $OUT/pmc_fm2 -- python3 tools/profile_frontend.py 500 1500 > /dev/null 2>> $OUT/pmc_fm.err || exit 1
{ echo "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 tools/profile_frontend.py 500 1500 (two passes); mean per dispatch";
  python3 tools/pmc_summary.py $OUT/pmc_fm1 $OUT/pmc_fm2 | grep hamming_mfma; } > $OUT/pmc_frontend500_matcher_${TAG}.txt
rm -rf $OUT/pmc_fm1 $OUT/pmc_fm2 $OUT/pmc_fm.err
