#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_step6.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/pytest_step6.log
tail -4 gpurun_out/r4/pytest_step6.log
if [ $rc != 0 ]; then exit 1; fi
python scripts/gpu_ab.py 128 5 build/librtow_prev.so - build/librtow_prev.so - > gpurun_out/r4/ab_step6.txt 2>&1
cat gpurun_out/r4/ab_step6.txt
