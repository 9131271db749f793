#!/bin/bash
# Round 4, call 8: epilogue straight from the reduce slots (no finalize pass); non-temporal loads of the compacted side's rows in the exact re-check.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4_pytest_gpu.log
timeout 600 python tools/ab_multi.py --p1 --rounds 12 r03=$A/libmsfm_match_r03.so tree nta=$A/libmsfm_nta.so > $OUT/r4_call8_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call8_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 r03=$A/libmsfm_match_r03.so tree nta=$A/libmsfm_nta.so > $OUT/r4_call8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call8.txt
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 10 r03=$A/libmsfm_match_r03.so tree nta=$A/libmsfm_nta.so > $OUT/r4_call8_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call8_u8.txt
MSFM_Q8=2 timeout 300 python tools/fuzz_routes.py 801 300 > $OUT/r4_call8_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/r4_call8_fuzz.txt
