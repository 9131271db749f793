#!/bin/bash
# Round 4, call 39: mixed sub-batches -- what one outlier image costs now; long job fuzz on the build with mixed sub-batches
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 300 python tools/mixed_store_ab.py 128 > $OUT/r4_mixed_store_ab.txt 2>&1; echo "rc=$?"; cat $OUT/r4_mixed_store_ab.txt
timeout 1500 python tools/fuzz_jobs.py 41 1500 > $OUT/r4_fuzz_jobs_e.txt 2>&1; echo "rc=$?"; tail -2 $OUT/r4_fuzz_jobs_e.txt
MSFM_Q8=2 timeout 1500 python tools/fuzz_jobs.py 42 1500 > $OUT/r4_fuzz_jobs_f.txt 2>&1; echo "rc=$?"; tail -2 $OUT/r4_fuzz_jobs_f.txt
