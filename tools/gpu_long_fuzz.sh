#!/bin/bash
# long seeded fuzz of the routes against the brute-force route on the final build: route Q forced on small images (mode by the
# twins' level / direct forced / refinement forced), sub-batches of 3 pairs with three sets in flight, default limits, the device
# RANSAC against its host twin
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
MSFM_Q8=2 timeout 900 python tools/fuzz_routes.py 701 2500 > $OUT/fuzz_long_q8.txt 2>&1; echo "q8 rc=$?"; tail -1 $OUT/fuzz_long_q8.txt
MSFM_Q8=2 MSFM_Q8_DIRECT=2 MSFM_MAX_PAIRS_PER_BATCH=3 timeout 900 python tools/fuzz_routes.py 702 1200 > $OUT/fuzz_long_q8_split.txt 2>&1; echo "q8 direct split rc=$?"; tail -1 $OUT/fuzz_long_q8_split.txt
MSFM_Q8=2 MSFM_Q8_DIRECT=0 MSFM_MAX_PAIRS_PER_BATCH=2 MSFM_IN_FLIGHT=2 timeout 900 python tools/fuzz_routes.py 703 800 > $OUT/fuzz_long_q8_refine_split.txt 2>&1; echo "q8 refine split rc=$?"; tail -1 $OUT/fuzz_long_q8_refine_split.txt
timeout 600 python tools/fuzz_routes.py 704 1500 > $OUT/fuzz_long_default.txt 2>&1; echo "default rc=$?"; tail -1 $OUT/fuzz_long_default.txt
timeout 600 python tools/fuzz_verify.py 2>&1 | tail -2
