#!/bin/bash
# Round 4, call 31: the job-level fuzz, long (new seeds)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python tools/fuzz_jobs.py 21 1500 > $OUT/r4_fuzz_jobs_c.txt 2>&1; echo "rc=$?"; tail -3 $OUT/r4_fuzz_jobs_c.txt
MSFM_Q8=2 timeout 1500 python tools/fuzz_jobs.py 22 1500 > $OUT/r4_fuzz_jobs_d.txt 2>&1; echo "rc=$?"; tail -3 $OUT/r4_fuzz_jobs_d.txt
