#!/bin/bash
# Round 4, call 7: the staged exact re-check (filtered fold, per-XCD cursor): parity tests first, then A/B against round 3.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A=tools/_ab
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r4_pytest_gpu.log
timeout 600 python tools/ab_multi.py --p1 --rounds 12 r03=$A/libmsfm_match_r03.so tree single=$A/libmsfm_single.so > $OUT/r4_exact_staged_p1.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_staged_p1.txt
timeout 600 python tools/ab_multi.py --rounds 12 r03=$A/libmsfm_match_r03.so tree single=$A/libmsfm_single.so > $OUT/r4_exact_staged.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_staged.txt
timeout 600 python tools/ab_multi.py --u8 --images 64 --rounds 10 r03=$A/libmsfm_match_r03.so tree single=$A/libmsfm_single.so > $OUT/r4_exact_staged_u8.txt 2>&1; echo "rc=$?"; cat $OUT/r4_exact_staged_u8.txt
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
cd /tmp; rm -rf $OUT/prof_p1 $OUT/pmc_fetch
MSFM_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_p1 -- $BENCH > $OUT/prof_p1.log 2>&1; echo "stats p1 rc=$?"
MSFM_PIPELINE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
cd $ROOT
DB=$(find $OUT/prof_p1 -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "MSFM_PIPELINE=1 $BENCH" > $OUT/r4_kernel_stats_p1.txt 2>&1; head -14 $OUT/r4_kernel_stats_p1.txt | cut -c1-60,150-230
python tools/pmc_summary.py $OUT/r4_pmc_traffic.json "sweep_i8_kernel<1>,sweep_kernel<3>,pf_exact_candidates_kernel" $OUT/pmc_fetch | grep -A3 exact | head -8
find $OUT/prof_p1 $OUT/pmc_fetch -type f -size +8M -delete
