set -e
export TMPDIR=/tmp
OUT=gpurun_out
stats() { local name=$1; shift; shift
  rm -rf $OUT/prof_tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tmp -- "$@" > $OUT/${name}.out 2> $OUT/prof_tmp.err
  find $OUT/prof_tmp -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/prof_tmp; }
stats frontend500_3a -- python3 tools/profile_frontend.py 500 1500
SLAMHIP_PROFILE_SHAPE=readme stats frontend500_readme_3a -- python3 tools/profile_frontend.py 500 1500
SLAMHIP_PROFILE_VARIANT=9 stats frontend500_g1_3a -- python3 tools/profile_frontend.py 500 1500
stats online_3a -- python3 tools/profile_online.py 1500
