#!/bin/bash
# Round 4, call 30: job-level fuzz (tools/fuzz_jobs.py): random stores under random cuts == the defaults == the C oracle
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1200 python tools/fuzz_jobs.py 11 150 > $OUT/r4_fuzz_jobs_a.txt 2>&1; echo "rc=$?"; tail -4 $OUT/r4_fuzz_jobs_a.txt
MSFM_Q8=2 timeout 1200 python tools/fuzz_jobs.py 12 150 > $OUT/r4_fuzz_jobs_b.txt 2>&1; echo "rc=$?"; tail -4 $OUT/r4_fuzz_jobs_b.txt
