#!/bin/bash
# Every profile this repository keeps under profiles/, from one GPU box (run from the repo root):
#   bash tools/profile_all.sh TAG        -> files under gpurun_out/, named *_TAG.*
# rocprofv3 kernel trace + stats per workload, the --pmc passes of the matcher benchmark
# (one run per counter group), the GPU test log and the benchmark line.
set -e
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out
stats() {   # stats NAME -- command...
  local name=$1; shift; shift
  rm -rf $OUT/prof_tmp
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tmp -- "$@" > $OUT/${name}_${TAG}.out 2> $OUT/prof_tmp.err
  find $OUT/prof_tmp -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats_${TAG}.csv \;
  rm -rf $OUT/prof_tmp
}
MARGS="--no-cpu-baseline --frames 0 --lc-keyframes 0 --sequence-frames 0 --no-sweep --no-pipelined"
stats bench_matcher8k -- python3 bench.py --steps 400 --warmup 20 $MARGS
SLAMHIP_MATCHER=popcount stats bench_matcher8k_popcount -- python3 bench.py --steps 400 --warmup 20 $MARGS
stats frontend500 -- python3 tools/profile_frontend.py 500 1500
SLAMHIP_PROFILE_SHAPE=readme stats frontend500_readme -- python3 tools/profile_frontend.py 500 1500
SLAMHIP_PROFILE_VARIANT=9 stats frontend500_one_group_per_wave -- python3 tools/profile_frontend.py 500 1500
stats online_pair -- python3 tools/profile_online.py 1500
stats frontend_envelope60 -- python3 tools/profile_frontend.py 60 2000 8000
stats loop_verify270 -- python3 tools/profile_loop_verify.py 270
stats l2_8k -- python3 tools/profile_l2.py 8192 256 10
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 20 --warmup 5 $MARGS > /dev/null 2>> $OUT/prof_tmp.err
done
{ echo "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 20 --warmup 5 $MARGS";
  echo "one run per counter group (FETCH_SIZE | WRITE_SIZE | 8 SQ counters | 8 SQ counters); mean per dispatch; 8192x8192x61B matcher";
  python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU $OUT/pmc_SQ_INSTS_MFMA; } > $OUT/pmc_matcher8k_${TAG}.txt
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU $OUT/pmc_SQ_INSTS_MFMA $OUT/prof_tmp.err
[ -x tools/mfma_mix_probe ] && timeout -k 10 60 tools/mfma_mix_probe > $OUT/mfma_mix_probe_${TAG}.log 2>&1
[ -x tools/bench_call ] && timeout -k 10 120 tools/bench_call 61 > $OUT/bench_call_${TAG}.log 2>&1
timeout -k 10 120 python3 tools/try_variants.py 8192 > $OUT/matcher_variants_${TAG}.log 2>&1
timeout -k 10 600 python3 -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu_${TAG}.log 2>&1
timeout -k 10 280 python3 bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err
