#!/bin/bash
# round 3, GPU call 1: sanity of the suite, then the what-if probes (each bounded; a killed step ends the script)
export TMPDIR=/tmp
out=gpurun_out/r3; mkdir -p $out
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; echo "[$(date +%T)] rc=$rc: $*" >> $out/progress.log
  if [ $rc = 124 ] || [ $rc = 137 ]; then echo "killed: $*" >> $out/progress.log; exit 1; fi; return 0; }
step 600 python -m pytest tests -m gpu -x -q > $out/pytest1.log 2>&1
step 300 python scripts/gpu_r3_probe.py reorder build/lib_reorder.so 128 3 sphere_scene > $out/whatif_reorder_sphere_scene.txt 2>&1
step 300 python scripts/gpu_r3_probe.py lanes build/lib_reorder_lanes.so 32 sphere_scene > $out/whatif_lanes_sphere_scene.txt 2>&1
step 300 python scripts/gpu_r3_probe.py reorder build/lib_reorder.so 64 3 cornell_box > $out/whatif_reorder_cornell_box.txt 2>&1
step 300 python scripts/gpu_r3_probe.py reorder build/lib_reorder.so 64 3 final_scene > $out/whatif_reorder_final_scene.txt 2>&1
step 300 python scripts/gpu_r3_probe.py reorder build/lib_reorder.so 128 3 pbr_sweep_scene > $out/whatif_reorder_pbr_sweep_scene.txt 2>&1
step 200 python scripts/gpu_r3_probe.py treemem build/lib_base.so 128 3 sphere_scene > $out/treemem_sphere_scene.txt 2>&1
RTOW_FORCE_GENERAL=1 step 200 python scripts/gpu_r3_probe.py treemem build/lib_base.so 128 3 sphere_scene > $out/treemem_sphere_scene_general.txt 2>&1
step 400 python scripts/fetch_calibration.py $out/fetch_calibration.json > $out/fetch_calibration.log 2>&1
echo done >> $out/progress.log
