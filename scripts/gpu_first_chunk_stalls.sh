#!/bin/bash
# How often does the FIRST frame of a fresh process run into a driver stall right after another process freed 100 GB, by chunk size
# of the pool?  (round 6)  Output: gpurun_out/r6/first_chunk_stalls.txt
out=gpurun_out/r6; mkdir -p $out; log=$out/first_chunk_stalls.txt; : > $log
P=scripts/micro/alloc_probe
run() { # run <label> <lib or empty>
  timeout -k 10 120 $P churn 100 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('  churn: hipMalloc %.0f ms, hipFree %.0f ms' % (d['hipMalloc_ms'], d['hipFree_ms']))" >> $log
  RTOW_GPU_LIB=$2 timeout -k 10 120 python bench.py --first-frame-child --config 2 $3 | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p = d['parts_ms']; q = d['first_render_parts_ms']
print('$1: first_frame %.0f ms = ctx %.0f + scene %.0f + upload %.0f + first render %.0f (%d slices, waited for the pool %.0f ms); slowest chunk %.0f ms' % (
      d['first_frame_ms'], p['rt_ctx_create'], p['scene_build_host'], p['rt_scene_upload'], p['first_rt_render'], d['first_render_slices'], q.get('of_which_waiting_for_the_pool', 0.0), q['pool_slowest_chunk_ms']))" >> $log
}
for k in 1 2 3 4 5 6; do
  run "128 MB chunks" ""
  run "512 MB chunks" build/librtow_chunk512.so
done
cat $log
