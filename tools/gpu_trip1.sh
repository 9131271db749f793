#!/bin/bash
# round 3, first GPU call: full GPU test suite (incl. the new exchange tests), config 4 in full, power trace, two-rank RCCL attempt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 900 python tools/config4_full.py > $OUT/config4_full.json 2> $OUT/config4_full.err; echo "config4 rc=$?"; head -c 1500 $OUT/config4_full.json; tail -3 $OUT/config4_full.err
bash tools/gpu_rccl_two_ranks.sh
bash tools/gpu_power_trace.sh
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 900 $OUT/bench.json
