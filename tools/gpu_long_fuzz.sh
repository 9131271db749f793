#!/bin/bash
# long seeded fuzz of the routes against the brute-force route (route Q forced on small images; default; fp16-only limits)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
MSFM_Q8=2 timeout 900 python tools/fuzz_routes.py 101 1500 > $OUT/fuzz_long_q8.txt 2>&1; echo "q8 rc=$?"; tail -2 $OUT/fuzz_long_q8.txt
MSFM_Q8=2 MSFM_MAX_PAIRS_PER_BATCH=3 timeout 900 python tools/fuzz_routes.py 102 700 > $OUT/fuzz_long_q8_split.txt 2>&1; echo "q8 split rc=$?"; tail -2 $OUT/fuzz_long_q8_split.txt
timeout 600 python tools/fuzz_routes.py 103 800 > $OUT/fuzz_long_default.txt 2>&1; echo "default rc=$?"; tail -2 $OUT/fuzz_long_default.txt
timeout 600 python tools/fuzz_oracle.py 2>&1 | tail -3
