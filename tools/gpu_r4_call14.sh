#!/bin/bash
# Round 4, call 14: which of v1's changes slows the PIPELINED byte job?  v4 = v1 with round 3's slot assignment, v5 = v1 with the 4-byte row loads in the exact re-check.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 12 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v4=$A/libmsfm_v4.so v5=$A/libmsfm_v5.so > $OUT/r4_call14_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call14_u8.txt
timeout 600 python tools/ab_multi.py --rounds 12 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v4=$A/libmsfm_v4.so v5=$A/libmsfm_v5.so > $OUT/r4_call14.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call14.txt
timeout 600 python tools/ab_multi.py --u8 --images 160 --rounds 6 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v4=$A/libmsfm_v4.so v5=$A/libmsfm_v5.so > $OUT/r4_call14_u8_160.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call14_u8_160.txt
