#!/bin/bash
# Round 4, call 10: hot-address atomics -- group totals by a kernel of their own; span cursor fetched per batch of spans.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 600 python tools/ab_multi.py --p1 --rounds 12 prev=$A/libmsfm_prev.so tree batch1=$A/libmsfm_batch1.so batch16=$A/libmsfm_batch16.so > $OUT/r4_call10_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call10_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 prev=$A/libmsfm_prev.so tree batch1=$A/libmsfm_batch1.so batch16=$A/libmsfm_batch16.so > $OUT/r4_call10.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call10.txt
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 10 prev=$A/libmsfm_prev.so tree batch1=$A/libmsfm_batch1.so > $OUT/r4_call10_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call10_u8.txt
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4_pytest_gpu.log
