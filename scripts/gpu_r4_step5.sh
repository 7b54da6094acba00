#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r4/pytest_step5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4/pytest_step5.log
tail -25 gpurun_out/r4/pytest_step5.log
