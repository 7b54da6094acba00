#!/bin/bash
# The measurements a round commits under profiles/roundN (run on the GPU box): bash scripts/gpu_round_profiles.sh profiles/round2
# bench lines of configs 2-5, the rocprofv3 kernel statistics of the same commands, PMC traffic and VALU passes of config 2,
# the per-dispatch PMC table and the VALU issue-rate microbenchmark.
export TMPDIR=/tmp
out=${1:-gpurun_out/profiles}; mkdir -p $out
phase=${2:-AB}   # A: microbenchmarks + PMC passes; B: bench lines, kernel statistics, per-dispatch PMC, scene table (one gpurun call each: 20 min limit)
py=$(python -c 'import sys; print(sys.executable)')
# Every GPU step is bounded, and a step that had to be killed ends the script: no further GPU work after a hang.
# (rocprofv3 keeps the interpreter directly behind `--`; the limit wraps rocprofv3 itself, which has not touched the GPU.)
step() { # step <seconds> <command...>
  local lim=$1; shift
  timeout -k 10 $lim "$@"; local rc=$?
  if [ $rc = 124 ] || [ $rc = 137 ]; then echo "step killed at its ${lim}s limit: $*" | tee -a $out/ABORTED.txt; exit 1; fi
  echo "[$(date +%T)] rc=$rc: $1 $2 $3 $4" >> $out/progress.log
  return 0
}
if [[ $phase == *A* ]]; then
step 120 ./scripts/micro/mul_rate $out/valu_peak.json > $out/mul_rate.txt 2>&1
# (the micro binaries are built by hand in the container, see the first lines of scripts/micro/*.hip, and travel with the snapshot)
[ -x ./scripts/micro/latency ] && step 120 ./scripts/micro/latency $out/latency.json > $out/latency.txt 2>&1
# The PMC passes first: bench.py quotes roofline.traffic / roofline.valu from profiles/round*/, so the files of THIS library
# have to be in place (in this box's copy of the repo; copy them into the tracked profiles/ afterwards) before the bench lines run.
prof=$(ls -d profiles/round* | sort -V | tail -1)
cp $out/valu_peak.json $prof/valu_peak.json
step 400 $py scripts/collect_traffic.py $out/traffic_config2.json > $out/traffic_config2.log 2>&1
step 400 $py scripts/collect_valu.py $out/valu_config2.json > $out/valu_config2.log 2>&1
step 400 $py scripts/collect_traffic.py $out/traffic_config4.json --config 4 --steps 1 --warmup 0 --timed-only > $out/traffic_config4.log 2>&1
step 400 $py scripts/collect_traffic.py $out/traffic_config5.json --config 5 --steps 1 --warmup 0 --timed-only > $out/traffic_config5.log 2>&1
cp $out/traffic_config2.json $out/traffic_config4.json $out/traffic_config5.json $out/valu_config2.json $prof/
fi
if [[ $phase == *B* ]]; then
prof=$(ls -d profiles/round* | sort -V | tail -1)
step 300 $py bench.py --steps 20 --warmup 5 > $out/bench_config2.json 2> $out/bench_config2.err
for c in 3 4 5; do step 300 $py bench.py --config $c --steps 3 --warmup 1 > $out/bench_config$c.json 2> $out/bench_config$c.err; done
for c in 2 4 5; do
  extra=""; [ $c != 2 ] && extra="--config $c"
  rm -rf $out/prof_tmp
  step 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_tmp -- $py bench.py $extra --steps 2 --warmup 1 --timed-only > $out/kernel_stats_config$c.log 2>&1
  find $out/prof_tmp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_config$c.csv
done
rm -rf $out/prof_tmp
step 400 $py scripts/pmc_probe.py $out/pmc_probe_tmp "" 128 > $out/pmc_per_dispatch_config2_128spp.txt 2>&1
rm -rf $out/pmc_probe_tmp gpurun_out/pmc_*
step 300 $py scripts/gpu_scene_table.py 64 $out/scene_table.json > $out/scene_table.txt 2>&1
fi
echo done
