#!/bin/bash
# Kernel time of every timing-experiment library built by tools/build_ablate.sh (GPU box, repo root).
for f in slam_amd/libslamhip.so slam_amd/libslamhip_abl*.so; do
  SLAMHIP_LIB_PATH=$PWD/$f timeout -k 10 60 python3 tools/time_abl.py 2>/dev/null | grep variant || exit 1
done
