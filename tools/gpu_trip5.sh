#!/bin/bash
# route Q (byte twins of float stores): correctness first, then A/B against the fp16 route on the bench job
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_q8.py -x -q > $OUT/pytest_q8.log 2>&1; echo "pytest q8 rc=$?"; tail -15 $OUT/pytest_q8.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
for round in 1 2; do
  for cfg in "0 1" "0 4" "1 1" "1 4"; do
    set -- $cfg
    MSFM_Q8=$1 MSFM_PIPELINE=$2 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 0 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('q8 $1 pipeline $2 round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step (frac %.3f of %s), sweep 1b %.2f ms (%.3f of the work), sweep 2 %.2f ms, sub-batches %d, checksum %s, sensitive rows %s' % (d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], r['frac'], r['unit'], r['route_q']['sweep1b_ms_per_step'], r['route_q']['sweep1b_work_fraction_of_sweep1'], r['sweep2']['ms_per_step'], d['sub_batches_per_step'], d['exchange_checksum'], d['order_sensitive_rows']))"
  done
done 2>&1 | tee $OUT/q8_ab.txt
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_stats
MSFM_PIPELINE=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --u8-images 0 --sustained-steps 0 > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
cd $ROOT
DB=$(find $OUT/prof_stats -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "MSFM_PIPELINE=1 bench.py --steps 3 --warmup 2" > $OUT/kernel_stats_q8.txt 2>&1; head -24 $OUT/kernel_stats_q8.txt | cut -c1-170
find $OUT/prof_stats -type f -size +8M -delete
