#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "final_scene or cornell or rectangles or translate or medium or random_scenes or hbm_resident or primary_candidate" > gpurun_out/r4/pytest_leafq.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/pytest_leafq.log
tail -6 gpurun_out/r4/pytest_leafq.log
if [ $rc = 124 ] || [ $rc = 137 ]; then exit 1; fi
for sc in final_scene cornell_box simple_light_scene; do
  RTOW_SCENE=$sc timeout -k 10 200 python scripts/gpu_ab.py 64 3 build/librtow_leafq0.so - build/librtow_leafq0.so - > gpurun_out/r4/ab_leafq_$sc.txt 2>&1 || exit 1
  cat gpurun_out/r4/ab_leafq_$sc.txt
done
