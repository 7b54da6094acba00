#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
python scripts/gpu_ab.py 128 3 - -@grid=1 -@grid_cell=1000 -@grid_cell=1200 -@grid_cell=1600 -@grid=1 - > gpurun_out/r4/ab_step2.txt 2>&1
cat gpurun_out/r4/ab_step2.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_step2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4/pytest_step2.log
tail -8 gpurun_out/r4/pytest_step2.log
