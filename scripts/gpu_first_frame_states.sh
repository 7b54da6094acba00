#!/bin/bash
# What the first frame of a fresh process costs, by the state of the device (round 6).  Runs scripts/micro/alloc_probe and
# bench.py's first-frame child in a quiet device, right after another process freed tens of GB, and beside a tenant that holds
# them.  Output: gpurun_out/r6/first_frame_states.txt.   usage: bash scripts/gpu_first_frame_states.sh
set -u
out=gpurun_out/r6
mkdir -p $out
log=$out/first_frame_states.txt
P=scripts/micro/alloc_probe
: > $log
say() { echo "== $*" | tee -a $log; }
child() { timeout -k 10 120 python bench.py --first-frame-child --config 2 $* 2>&1 | grep '^{' >> $log; }

say "1. quiet device: allocation sizes"
timeout -k 10 120 $P probe quiet >> $log 2>&1
say "2. quiet device: first-frame child (with rt_prepare twice, without once)"
child
child
child --no-prepare
say "3. right after a process that allocated, touched and freed 100 GB: small requests first, then sizes"
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
timeout -k 10 120 $P small after_churn_100 6 64 >> $log 2>&1
say "3b. churn again, then the size series"
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
timeout -k 10 120 $P probe after_churn_100 >> $log 2>&1
say "4. right after churn 100 GB: first-frame child, with and without rt_prepare"
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
child
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
child --no-prepare
say "5. after 3 x 53 GB churn (three frames' worth of buffers freed): first-frame child"
timeout -k 10 120 $P churn 53 3 >> $log 2>&1
child
say "6. settled (8 s idle): first-frame child"
sleep 8
child
say "7. beside a tenant holding 60 GB (started 4 s before): sizes, then first-frame child"
timeout -k 10 90 $P hold 60 40 >> $log 2>&1 &
holder=$!
sleep 4
timeout -k 10 120 $P probe beside_tenant_60 >> $log 2>&1
child
wait $holder
say "8. VMM: 67 GB in 1 GB chunks, quiet (8 s idle) and right after churn"
sleep 8
timeout -k 10 120 $P vmm quiet 1024 67 >> $log 2>&1
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
timeout -k 10 120 $P vmm after_churn_100 1024 67 >> $log 2>&1
say "8b. VMM with 256 MB chunks right after churn"
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
timeout -k 10 120 $P vmm after_churn_100 256 67 >> $log 2>&1
say "9. a helper thread allocates 53 GB while the main thread streams: quiet, then after churn"
sleep 8
timeout -k 10 120 $P concurrent quiet 53 >> $log 2>&1
timeout -k 10 120 $P churn 100 1 >> $log 2>&1
timeout -k 10 120 $P concurrent after_churn_100 53 >> $log 2>&1
say done
