#!/bin/bash
# round-2 first GPU call: full GPU test suite, bench line, A/B of the sweep variants
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.json; tail -5 $OUT/bench.err
timeout 600 bash tools/variant_bench.sh "base=" "noprio=-DMSFM_SWEEP_NOPRIO" > $OUT/variants.txt 2>&1; cat $OUT/variants.txt
