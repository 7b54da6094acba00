#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "grid or sphere_scene or config1 or config2 or sharding or slices or other_scenes or roulette or pbr" > gpurun_out/r4/pytest_prelim.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r4/pytest_prelim.log
tail -6 gpurun_out/r4/pytest_prelim.log
if [ $rc = 124 ] || [ $rc = 137 ]; then exit 1; fi
python scripts/gpu_ab.py 128 3 build/librtow_prev.so - build/librtow_prev.so - > gpurun_out/r4/ab_prelim_one_chain.txt 2>&1
cat gpurun_out/r4/ab_prelim_one_chain.txt
RTOW_AB_FLAGS=0 python scripts/gpu_ab.py 256 5 build/librtow_prev.so - build/librtow_prev.so - > gpurun_out/r4/ab_prelim_frame.txt 2>&1
cat gpurun_out/r4/ab_prelim_frame.txt
RTOW_SCENE=pbr_sweep_scene RTOW_AB_FLAGS=0 python scripts/gpu_ab.py 256 3 build/librtow_prev.so - build/librtow_prev.so - > gpurun_out/r4/ab_prelim_frame_pbr.txt 2>&1
cat gpurun_out/r4/ab_prelim_frame_pbr.txt
