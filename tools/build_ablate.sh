#!/bin/bash
# Builds timing-experiment copies of the library: slam_amd/libslamhip_ablN.so for every N given
# (hamming_mfma.hip compiled with -DSLAMHIP_ABLATE=N, the other objects as they are).
#   tools/build_ablate.sh 1 2 3 ...      (run after make -C slam_amd/csrc)
set -e
cd "$(dirname "$0")/../slam_amd/csrc"
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DSLAMHIP_ABLATE=$N $EXTRA -c hamming_mfma.hip -o /tmp/hamming_mfma_abl$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_abl$N.so hamming.o /tmp/hamming_mfma_abl$N.o l2.o geometry.o ransac.o p3p.o refine.o pipeline.o records.o
done
