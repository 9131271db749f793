#!/bin/bash
# Round 4, call 42: mixed sub-batches with the second route's own item list: what one outlier image costs; tests; job fuzz
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 300 python tools/mixed_store_ab.py 128 > $OUT/r4_mixed_store_ab.txt 2>&1; echo "rc=$?"; cat $OUT/r4_mixed_store_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/r4_mixed_pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/r4_mixed_pytest.txt
timeout 900 python tools/fuzz_jobs.py 51 600 > $OUT/r4_fuzz_jobs_g.txt 2>&1; echo "rc=$?"; tail -1 $OUT/r4_fuzz_jobs_g.txt
MSFM_Q8=2 timeout 900 python tools/fuzz_jobs.py 52 1000 > $OUT/r4_fuzz_jobs_h.txt 2>&1; echo "rc=$?"; tail -1 $OUT/r4_fuzz_jobs_h.txt
