#!/bin/bash
# Round 4, call 5: where should a candidate be evaluated?  Variants of the exact re-check's mapping (span length, XCD-local or
# not) and sweep 1 with / without the next-item record prefetch, all against round 3's library on one box; then the GPU tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 600 python tools/ab_multi.py --p1 --rounds 12 r03=$A/libmsfm_match_r03.so tree noprefetch=$A/libmsfm_noprefetch.so span16=$A/libmsfm_span16.so span64=$A/libmsfm_span64.so span1024=$A/libmsfm_span1024.so global=$A/libmsfm_global.so global16=$A/libmsfm_global16.so > $OUT/r4_exact_variants_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_variants_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 r03=$A/libmsfm_match_r03.so tree noprefetch=$A/libmsfm_noprefetch.so span16=$A/libmsfm_span16.so global16=$A/libmsfm_global16.so > $OUT/r4_exact_variants.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_variants.txt
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 10 r03=$A/libmsfm_match_r03.so tree noprefetch=$A/libmsfm_noprefetch.so span16=$A/libmsfm_span16.so global16=$A/libmsfm_global16.so > $OUT/r4_exact_variants_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_variants_u8.txt
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4_pytest_gpu.log
