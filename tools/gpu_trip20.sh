#!/bin/bash
# full GPU suite + fuzz + bench on the new defaults (one-barrier integer sweep, direct route-Q thresholds, 3 sets x 6 tapered parts)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 501 600 > $OUT/fuzz_a.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_a.txt
timeout 400 python tools/fuzz_routes.py 502 600 > $OUT/fuzz_b.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_b.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.json
