#!/bin/bash
# usage (on the GPU box): bash scripts_gpu_sweep.sh
export TMPDIR=/tmp
for s in 8 16 32 64 128 256; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp-slice $s 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('slice', $s, 'Mray/s', j['value'], 'ms', j['ms_per_step'], 'trace frac', j['roofline']['frac'], 'launch_us', j['roofline']['avg_launch_us'])"
done
