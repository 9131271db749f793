#!/bin/bash
# round 3, second GPU call: the pipelined match_pairs (two streams / scratch sets) + the order-invariance certificate
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
# pipeline depth A/B on the bench job, alternated twice on this box
for round in 1 2; do
  for p in 1 2 4 6 8; do
    MSFM_PIPELINE=$p timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 0 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('pipeline %s round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step (frac %.3f), step / sweep 1 %.3f, sub-batches %d, sweep 2 %.2f ms, checksum %s, sensitive rows %s' % (d['pipeline_env'], d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], r['frac'], r['step_over_sweep1'], d['sub_batches_per_step'], r['sweep2']['ms_per_step'], d['exchange_checksum'], d['order_sensitive_rows']))"
  done
done 2>&1 | tee $OUT/pipeline_ab.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 700 $OUT/bench.json
timeout 900 python tools/config4_full.py --int-oracle-pairs 1 > $OUT/config4_full.json 2> $OUT/config4_full.err; echo "config4 rc=$?"; head -c 1300 $OUT/config4_full.json; tail -3 $OUT/config4_full.err
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_stats
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --u8-images 0 --sustained-steps 0 > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
cd $ROOT
DB=$(find $OUT/prof_stats -name '*.db' | head -1)
python tools/rocprof_summary.py "$DB" "bench.py --steps 3 --warmup 2" > $OUT/kernel_stats.txt 2>&1; head -14 $OUT/kernel_stats.txt | cut -c1-170
python tools/step_timeline.py "$DB" > $OUT/step_timeline.txt 2>&1; tail -5 $OUT/step_timeline.txt
find $OUT/prof_stats -type f -size +8M -delete
