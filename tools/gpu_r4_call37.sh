#!/bin/bash
# Round 4, call 37: mixed sub-batches (twins' pairs on the integer cores, the rest on the fp16 cores in ONE sub-batch): GPU tests, job fuzz, route fuzz
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/r4_mixed_pytest.txt 2>&1; echo "rc=$?"; tail -4 $OUT/r4_mixed_pytest.txt
timeout 600 python tools/fuzz_jobs.py 31 400 > $OUT/r4_mixed_fuzz_jobs_a.txt 2>&1; echo "rc=$?"; tail -2 $OUT/r4_mixed_fuzz_jobs_a.txt
MSFM_Q8=2 timeout 600 python tools/fuzz_jobs.py 32 400 > $OUT/r4_mixed_fuzz_jobs_b.txt 2>&1; echo "rc=$?"; tail -2 $OUT/r4_mixed_fuzz_jobs_b.txt
MSFM_Q8=2 timeout 600 python tools/fuzz_routes.py 1401 800 > $OUT/r4_mixed_fuzz_routes.txt 2>&1; echo "rc=$?"; tail -1 $OUT/r4_mixed_fuzz_routes.txt
