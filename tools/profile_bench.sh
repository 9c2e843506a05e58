#!/bin/bash
# rocprofv3 passes over the matcher benchmark (run on the GPU box from the repo root):
# kernel trace + stats, then one --pmc pass per counter group.  Output under gpurun_out/.
set -e
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --frames 0 --lc-keyframes 0 --no-sweep"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 400 --warmup 20 $ARGS > gpurun_out/prof_bench.json
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 20 --warmup 5 $ARGS > /dev/null
done
python3 tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_INSTS_VALU > gpurun_out/pmc_matcher8k.txt
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/bench_matcher8k_kernel_stats.csv \;
