#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
B=build
python scripts/gpu_ab.py 128 3 - $B/librtow_nopbr4.so $B/librtow_nopbr5.so $B/librtow_nopbr6.so $B/librtow_w5.so $B/librtow_w6.so - > gpurun_out/r4/ab_shade_waves_one_chain.txt 2>&1
cat gpurun_out/r4/ab_shade_waves_one_chain.txt
RTOW_AB_FLAGS=0 python scripts/gpu_ab.py 128 5 - $B/librtow_nopbr4.so $B/librtow_nopbr5.so $B/librtow_nopbr6.so $B/librtow_w5.so $B/librtow_w6.so - > gpurun_out/r4/ab_shade_waves_frame.txt 2>&1
cat gpurun_out/r4/ab_shade_waves_frame.txt
