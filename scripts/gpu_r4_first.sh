#!/bin/bash
# round 4, first GPU call: the grid tests, then the grid against the tree on config-2 content (one chain, per-kernel times;
# then the production two-chain frame)
set -o pipefail
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -k "grid or adversarial or abi" > gpurun_out/r4/pytest_grid.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4/pytest_grid.log
tail -5 gpurun_out/r4/pytest_grid.log
python scripts/gpu_ab.py 128 3 -@grid=1 - -@grid_cell=700 -@grid_cell=1400 -@grid_cell=2000 -@grid=1 > gpurun_out/r4/ab_grid_first.txt 2>&1
cat gpurun_out/r4/ab_grid_first.txt
RTOW_AB_FLAGS=0 python scripts/gpu_ab.py 256 3 -@grid=1 - -@grid_cell=1400 -@grid=1 > gpurun_out/r4/ab_grid_first_frame.txt 2>&1
cat gpurun_out/r4/ab_grid_first_frame.txt
