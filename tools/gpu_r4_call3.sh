#!/bin/bash
# Round 4, call 3: balanced XCD-local exact re-check: GPU tests, A/B against round 3's library, unpipelined kernel stats + traffic.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/r4_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r4_pytest_gpu.log
timeout 300 python tools/ab_libs.py tools/_ab/libmsfm_match_r03.so > $OUT/r4_ab_f32.txt 2>&1; echo "ab f32 rc=$?"; cat $OUT/r4_ab_f32.txt | tail -3
timeout 300 python tools/ab_libs.py tools/_ab/libmsfm_match_r03.so p1 > $OUT/r4_ab_f32_p1.txt 2>&1; echo "ab f32 p1 rc=$?"; cat $OUT/r4_ab_f32_p1.txt | tail -3
timeout 300 python tools/ab_libs.py tools/_ab/libmsfm_match_r03.so u8 > $OUT/r4_ab_u8.txt 2>&1; echo "ab u8 rc=$?"; cat $OUT/r4_ab_u8.txt | tail -3
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
cd /tmp; rm -rf $OUT/prof_p1 $OUT/pmc_fetch $OUT/pmc_write
MSFM_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_p1 -- $BENCH > $OUT/prof_p1.log 2>&1; echo "stats p1 rc=$?"
MSFM_PIPELINE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
cd $ROOT
DB=$(find $OUT/prof_p1 -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "MSFM_PIPELINE=1 $BENCH" > $OUT/r4_kernel_stats_p1.txt 2>&1; head -16 $OUT/r4_kernel_stats_p1.txt | cut -c1-60,150-230
python tools/pmc_summary.py $OUT/r4_pmc_traffic.json "sweep_i8_kernel<1>,sweep_kernel<3>,pf_prune_q8_kernel,pf_assign_kernel,pf_exact_candidates_kernel,pf_finalize_kernel,epilogue_kernel,fill_segs_kernel" $OUT/pmc_fetch > /dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_pmc_traffic.json"))
for k,v in d.items(): print(k, {c:(round(x.get("per_launch_KB_mean",0)/1e6,3),x["launches"]) for c,x in v.items()})
PY
find $OUT/prof_p1 $OUT/pmc_fetch $OUT/pmc_write -type f -size +8M -delete
