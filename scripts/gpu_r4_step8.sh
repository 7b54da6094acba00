#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_step8.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/pytest_step8.log; tail -3 gpurun_out/r4/pytest_step8.log
[ $rc = 0 ] || exit 1
RTOW_AB_FLAGS=0 python scripts/gpu_ab.py 256 5 build/librtow_prev3.so - build/librtow_prev3.so - > gpurun_out/r4/ab_step8_frame.txt 2>&1
cat gpurun_out/r4/ab_step8_frame.txt
