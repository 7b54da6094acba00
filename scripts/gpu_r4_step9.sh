#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_step9.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/pytest_step9.log; tail -3 gpurun_out/r4/pytest_step9.log
[ $rc = 0 ] || exit 1
python scripts/gpu_ab.py 128 5 build/librtow_prev4.so - build/librtow_prev4.so - > gpurun_out/r4/ab_step9.txt 2>&1
cat gpurun_out/r4/ab_step9.txt
for sc in earth_env_scene cornell_box; do RTOW_SCENE=$sc python scripts/gpu_ab.py 64 3 build/librtow_prev4.so - build/librtow_prev4.so - > gpurun_out/r4/ab_step9_$sc.txt 2>&1; cat gpurun_out/r4/ab_step9_$sc.txt; done
