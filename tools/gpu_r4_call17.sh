#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/r4_pytest_gpu.log
