#!/bin/bash
# Round 4, call 15: exact re-check with DPP row shifts instead of LDS shuffles (tree) against v1 and the build before this series (prev); parity first.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4_pytest_gpu.log
timeout 600 python tools/ab_multi.py --p1 --rounds 12 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so tree > $OUT/r4_call15_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call15_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so tree > $OUT/r4_call15.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call15.txt
timeout 600 python tools/ab_multi.py --u8 --images 160 --rounds 6 prev=$A/libmsfm_prev.so v1=$A/libmsfm_v1.so tree > $OUT/r4_call15_u8_160.txt 2>&1; echo "rc=$?"; cat $OUT/r4_call15_u8_160.txt
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
cd /tmp; rm -rf $OUT/prof_p1
MSFM_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_p1 -- $BENCH > $OUT/prof_p1.log 2>&1; echo "stats p1 rc=$?"
cd $ROOT
DB=$(find $OUT/prof_p1 -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "MSFM_PIPELINE=1 $BENCH" > $OUT/r4_kernel_stats_p1.txt 2>&1; head -12 $OUT/r4_kernel_stats_p1.txt | cut -c1-60,150-215
MSFM_Q8=2 timeout 300 python tools/fuzz_routes.py 951 300 > $OUT/r4_call15_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/r4_call15_fuzz.txt
find $OUT/prof_p1 -type f -size +8M -delete
