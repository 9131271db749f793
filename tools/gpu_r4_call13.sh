#!/bin/bash
# Round 4, call 13: v1 = permuted rows with contiguous 16-byte loads + prune / thresholds / assign with loads up front; v2 = v1 + the sweep-1
# partials read as a stream (non-temporal); v3 = v2 with round 3's thresholds kernel (the byte job got slower with the new one).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 600 python tools/ab_multi.py --p1 --rounds 12 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v2=$A/libmsfm_v2.so v3=$A/libmsfm_v3.so > $OUT/r4_call13_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call13_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v2=$A/libmsfm_v2.so v3=$A/libmsfm_v3.so > $OUT/r4_call13.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call13.txt
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 10 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v2=$A/libmsfm_v2.so v3=$A/libmsfm_v3.so > $OUT/r4_call13_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call13_u8.txt
timeout 600 python tools/ab_multi.py --u8 --p1 --images 64 --rounds 10 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so v2=$A/libmsfm_v2.so v3=$A/libmsfm_v3.so > $OUT/r4_call13_u8_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call13_u8_p1.txt
