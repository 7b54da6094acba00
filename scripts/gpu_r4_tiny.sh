#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_tiny.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/pytest_tiny.log; tail -3 gpurun_out/r4/pytest_tiny.log
[ $rc = 0 ] || exit 1
for sc in cornell_box simple_light_scene earth_env_scene test_sphere; do
  RTOW_SCENE=$sc python scripts/gpu_ab.py 64 3 build/librtow_prev.so - build/librtow_prev.so - > gpurun_out/r4/ab_tiny_$sc.txt 2>&1
  echo $sc; cat gpurun_out/r4/ab_tiny_$sc.txt
done
