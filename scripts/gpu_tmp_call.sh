#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3; mkdir -p $out; rm -f $out/ab_tiles.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest18.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest18.log
for sc in sphere_scene pbr_sweep_scene cornell_box final_scene; do
RTOW_SCENE=$sc RTOW_AB_DEPTHS=2 timeout -k 10 300 python scripts/gpu_ab.py 128 5 build/lib_rowmajor.so build/lib_tiles.so build/lib_rowmajor.so build/lib_tiles.so 2>&1 | tee -a $out/ab_tiles.txt
done
RTOW_SWEEP_ONE_CTX=1 RTOW_SWEEP_FLAGS=0 timeout -k 10 400 python scripts/gpu_env_sweep.py sphere_scene 256 7 RTOW_ROW_MAJOR 1 - 1 - 2>&1 | tee -a $out/ab_tiles.txt
