#!/bin/bash
# rocprofv3 kernel statistics of the 500-frame pass (13 passes) -> gpurun_out/frontend500_kstats.csv
export TMPDIR=/tmp
rm -rf gpurun_out/fe_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fe_prof -- python3 tools/profile_frontend.py ${1:-500} ${2:-1500} > gpurun_out/fe_prof.log 2>&1
find gpurun_out/fe_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/frontend500_kstats.csv \;
rm -rf gpurun_out/fe_prof
tail -1 gpurun_out/fe_prof.log
