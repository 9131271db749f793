#!/bin/bash
# second-best reduction fused into the exact re-check: parity (full suite + fuzz) and unpipelined kernel times
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 801 800 > $OUT/fuzz_a.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_a.txt
timeout 400 python tools/fuzz_routes.py 802 800 > $OUT/fuzz_b.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_b.txt
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0 --no-solo"
cd /tmp; rm -rf $OUT/prof_stats_p1
MSFM_PIPELINE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_p1 -- $BENCH > $OUT/prof_stats_p1.log 2>&1; echo "stats p1 rc=$?"
cd $ROOT
DB1=$(ls -t $(find $OUT/prof_stats_p1 -name '*.db') | head -1)
python tools/rocprof_summary.py "$DB1" "MSFM_PIPELINE=1 $BENCH" > $OUT/kernel_stats_p1.txt 2>&1; head -14 $OUT/kernel_stats_p1.txt | cut -c1-150
find $OUT/prof_stats_p1 -type f -size +8M -delete
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --sustained-steps 0 --u8-images 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('%.2f ms per step %.3e/s | in-region frac %.3f | solo frac %.3f, unpipelined step %.2f ms' % (d['ms_per_step'], d['value'], r['frac'], r['solo']['frac'], r['solo']['ms_per_step_unpipelined']))"; done
