#!/bin/bash
# Two builds of the library side by side on one GPU box (alternating processes, so that both see
# the same chip and clocks): matcher step + top-2 kernel, 500-frame front-end pass.
#   tools/ab_compare.sh ab/libslamhip_prev.so   (the tree's own library is the other side)
PREV=${1:-ab/libslamhip_prev.so}
for rnd in 1 2 3; do
  for side in prev new; do
    if [ $side = prev ]; then export SLAMHIP_LIB_PATH=$PWD/$PREV; else unset SLAMHIP_LIB_PATH; fi
    m=$(timeout -k 10 120 python3 bench.py --steps 2000 --warmup 100 --no-sweep --frames 0 --sequence-frames 0 --lc-keyframes 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('step %.2f us  kernel %.2f us (b2b %.2f)' % (d['ms_per_step']*1e3, r['kernel_us'], r['kernel_us_back_to_back']))") || exit 1
    f=$(timeout -k 10 120 python3 tools/profile_frontend.py 500 1500 2>/dev/null | grep -o "[0-9.]* ms per pass") || exit 1
    e=$(timeout -k 10 120 python3 tools/profile_frontend.py 60 2000 8000 2>/dev/null | grep -o "[0-9.]* ms per pass") || exit 1
    echo "$side: $m | 500x1500 $f | 60x2000-8000 $e"
  done
done
