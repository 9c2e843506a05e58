#!/bin/bash
# Builds slam_amd/libslamhip_stamps.so: pipeline.hip with -DSLAMHIP_STAMPS (phase clocks of the fused
# small-call launches), the other objects as they are.  Run after make -C slam_amd/csrc.
set -e
cd "$(dirname "$0")/../slam_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -DSLAMHIP_STAMPS -c pipeline.hip -o /tmp/pipeline_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_stamps.so hamming.o hamming_mfma.o l2.o geometry.o ransac.o p3p.o refine.o /tmp/pipeline_stamps.o records.o
