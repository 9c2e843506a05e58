#!/bin/bash
# Counters of the front-end's consensus-counting launches (500 frames x 1500 features), one rocprofv3 --pmc pass,
# per dispatch: bash tools/pmc_frontend_count.sh TAG  ->  gpurun_out/pmc_frontend500_ransac_TAG.txt
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/pmc_fe
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/pmc_fe -- python3 tools/profile_frontend.py 500 1500 > /dev/null 2> $OUT/pmc_fe.err
{ echo "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -- python3 tools/profile_frontend.py 500 1500";
  echo "ransac_count_rectified_kernel, the first nine dispatches (three passes x rounds 1-3):";
  python3 tools/pmc_rows.py $OUT/pmc_fe ransac_count_rectified 9;
  echo "kernel durations of the same run (ns): ";
  python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_fe/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "ransac_count_rectified" in r["Kernel_Name"]]
    print([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[:9]])
PY
} > $OUT/pmc_frontend500_ransac_${TAG}.txt
rm -rf $OUT/pmc_fe $OUT/pmc_fe.err
