#!/bin/bash
# rocprofv3 kernel statistics of the 270-keyframe pair verification -> gpurun_out/loop_verify_kstats.csv
export TMPDIR=/tmp
rm -rf gpurun_out/lv_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lv_prof -- python3 tools/profile_loop_verify.py 270 > gpurun_out/lv_prof.log 2>&1
find gpurun_out/lv_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/loop_verify_kstats.csv \;
rm -rf gpurun_out/lv_prof
tail -1 gpurun_out/lv_prof.log
