#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
python scripts/gpu_ab.py 128 3 - build/librtow_w1t1.so -@grid=1 - > gpurun_out/r4/ab_step3.txt 2>&1
cat gpurun_out/r4/ab_step3.txt
python scripts/gpu_lane_stats.py build/librtow_lanes.so 32 sphere_scene > gpurun_out/r4/lanes_grid2.txt 2>&1
head -6 gpurun_out/r4/lanes_grid2.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_step3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4/pytest_step3.log
tail -8 gpurun_out/r4/pytest_step3.log
