#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/pytest_step4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4/pytest_step4.log
tail -8 gpurun_out/r4/pytest_step4.log
RTOW_SCENE=pbr_sweep_scene python scripts/gpu_ab.py 128 3 - -@grid=1 -@grid_cell=1000 -@grid_cell=2000 - > gpurun_out/r4/ab_pbr.txt 2>&1
cat gpurun_out/r4/ab_pbr.txt
RTOW_AB_FLAGS=0 python scripts/gpu_ab.py 256 3 - -@grid=1 - -@grid=1 > gpurun_out/r4/ab_frame_step4.txt 2>&1
cat gpurun_out/r4/ab_frame_step4.txt
