#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 600 python tools/ab_multi.py --p1 --rounds 12 r03=$A/libmsfm_match_r03.so tree nta=$A/libmsfm_nta.so > $OUT/r4_call8_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call8_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 r03=$A/libmsfm_match_r03.so tree nta=$A/libmsfm_nta.so > $OUT/r4_call8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call8.txt
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 10 r03=$A/libmsfm_match_r03.so tree nta=$A/libmsfm_nta.so > $OUT/r4_call8_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call8_u8.txt
