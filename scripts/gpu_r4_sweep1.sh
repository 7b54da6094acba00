#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
B=build
# (at most 8 contexts of 128 spp per process: each holds 27 GB of work buffers)
python scripts/gpu_ab.py 128 3 -@grid_cell=1400 $B/librtow_refill32.so@grid_cell=1400 $B/librtow_refill24.so@grid_cell=1400 \
  $B/librtow_test1.so@grid_cell=1400 $B/librtow_test40.so@grid_cell=1400 $B/librtow_walk1.so@grid_cell=1400 $B/librtow_walk4.so@grid_cell=1400 -@grid_cell=1400 > gpurun_out/r4/sweep1a.txt 2>&1
cat gpurun_out/r4/sweep1a.txt
python scripts/gpu_ab.py 128 3 -@grid_cell=1400 $B/librtow_big27.so@grid_cell=1400 $B/librtow_big200.so@grid_cell=1400 -@grid_cell=1200 -@grid_cell=1600 -@grid_cell=1800 -@grid=1 -@grid_cell=1400 > gpurun_out/r4/sweep1b.txt 2>&1
cat gpurun_out/r4/sweep1b.txt
