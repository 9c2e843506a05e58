#!/bin/bash
# rocprofv3 kernel-trace statistics of a few hundred matcher calls of one variant (GPU box, repo root):
#   tools/kernel_stats.sh VARIANT [n] [desc_bytes] [splits]  -> gpurun_out/kstats_vVARIANT.csv
export TMPDIR=/tmp
V=$1; N=${2:-8192}; D=${3:-61}; S=${4:-0}
rm -rf gpurun_out/kstats_$V
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_$V -- python3 tools/run_matcher_once.py $V $N $D $S 400 > /dev/null 2>&1
find gpurun_out/kstats_$V -name "*kernel_stats.csv" -exec cp {} gpurun_out/kstats_v$V.csv \;
cat gpurun_out/kstats_v$V.csv | cut -d, -f1-8 | sed 's/(anonymous namespace):://; s/void //' | cut -c1-160
