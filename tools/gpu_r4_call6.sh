#!/bin/bash
# Round 4, call 6: what makes the new exact re-check slower than round 3's although it fetches less?  Timing-only variants
# (no stores / no atomics / more or fewer workgroups); sweep 1 with four times the L2 -> LDS traffic.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 600 python tools/ab_multi.py --p1 --nocheck --rounds 12 r03=$A/libmsfm_match_r03.so tree nostore=$A/libmsfm_nostore.so noatomic=$A/libmsfm_noatomic.so nostoreatomic=$A/libmsfm_nostoreatomic.so wg64=$A/libmsfm_wg64.so wg2=$A/libmsfm_wg2.so g16wg64=$A/libmsfm_g16wg64.so > $OUT/r4_exact_hyp_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_hyp_p1.txt
timeout 600 python tools/ab_multi.py --p1 --rounds 12 tree dma4=$A/libmsfm_dma4.so noprefetch=$A/libmsfm_noprefetch.so > $OUT/r4_i8_dma_x4.txt 2>&1; echo "rc=$?"; cat $OUT/r4_i8_dma_x4.txt
timeout 600 python tools/ab_multi.py --p1 --u8 --images 64 --rounds 10 tree dma4=$A/libmsfm_dma4.so noprefetch=$A/libmsfm_noprefetch.so > $OUT/r4_i8_dma_x4_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_i8_dma_x4_u8.txt
