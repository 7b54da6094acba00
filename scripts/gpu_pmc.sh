#!/bin/bash
# PMC passes for the trace kernel (each counter group in its own run; --kernel-trace only).
# usage on the GPU box: bash scripts/gpu_pmc.sh <outdir> [bench args]
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc}; shift
args=${@:---steps 1 --warmup 0 --no-cpu-baseline --spp 32}
mkdir -p $out
py=$(python -c 'import sys; print(sys.executable)')  # the interpreter binary itself: no PATH shim behind rocprofv3's `--`
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- $py bench.py $args > $out/g$i.log 2>&1
  echo "group $i ($grp) exit $?"
done
