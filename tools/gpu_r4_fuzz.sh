#!/bin/bash
# Round 4: long seeded fuzz of the final build (routes against the brute-force route, device against the CPU oracle, device RANSAC against its host twin)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4fuzz; rm -rf $OUT; mkdir -p $OUT
( time MSFM_Q8=2 timeout 420 python tools/fuzz_routes.py 1601 1500 ) > $OUT/fuzz_q8.txt 2>&1; echo "fuzz q8 rc=$?"; tail -5 $OUT/fuzz_q8.txt
( time MSFM_Q8=2 MSFM_Q8_DIRECT=2 timeout 300 python tools/fuzz_routes.py 1602 800 ) > $OUT/fuzz_q8_direct.txt 2>&1; echo "fuzz q8 direct rc=$?"; tail -5 $OUT/fuzz_q8_direct.txt
( time MSFM_Q8=2 MSFM_Q8_DIRECT=0 timeout 300 python tools/fuzz_routes.py 1603 600 ) > $OUT/fuzz_q8_refine.txt 2>&1; echo "fuzz q8 refine rc=$?"; tail -5 $OUT/fuzz_q8_refine.txt
( time timeout 420 python tools/fuzz_routes.py 1604 1500 ) > $OUT/fuzz_default.txt 2>&1; echo "fuzz rc=$?"; tail -5 $OUT/fuzz_default.txt
( time timeout 300 python tools/fuzz_oracle.py 1605 400 ) > $OUT/fuzz_oracle.txt 2>&1; echo "fuzz oracle rc=$?"; tail -5 $OUT/fuzz_oracle.txt
( time timeout 200 python tools/fuzz_verify.py 1606 200 ) > $OUT/fuzz_verify.txt 2>&1; echo "fuzz verify rc=$?"; tail -5 $OUT/fuzz_verify.txt
