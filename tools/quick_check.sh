#!/bin/bash
# Quick GPU round: pose / front-end tests, then the timings that follow a kernel change.
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_pose.py tests/test_gpu_geometry.py tests/test_gpu_frontend.py -x -q --timeout 120 --timeout-method thread > gpurun_out/quick_tests.log 2>&1
tail -3 gpurun_out/quick_tests.log
timeout -k 10 60 python3 tools/time_refine.py 2>&1 | grep -E "^ +(1|499) problems" || exit 1
timeout -k 10 60 python3 tools/profile_online.py 1500 2>&1 | grep pair || exit 1
timeout -k 10 60 python3 tools/profile_online.py 1500 14=0 2>&1 | grep pair || exit 1
timeout -k 10 60 python3 tools/profile_online.py 3000 2>&1 | grep pair || exit 1
timeout -k 10 90 python3 tools/profile_frontend.py 500 1500 2>&1 | grep "per pass" || exit 1
SLAMHIP_PROFILE_SHAPE=readme timeout -k 10 90 python3 tools/profile_frontend.py 500 1500 2>&1 | grep "per pass" || exit 1
