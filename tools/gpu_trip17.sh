#!/bin/bash
# three sets in flight with S1(k+2) ordered behind S2(k)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for round in 1 2; do
  for cfg in "2 4" "3 4" "3 6" "3 8" "3 5"; do
    set -- $cfg
    MSFM_IN_FLIGHT=$1 MSFM_PIPELINE=$2 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('in flight $1 pipeline $2 round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step (frac %.3f), checksum %s | u8 job %.2f ms per step %.3e/s sweep-1 frac %.3f' % (d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], r['frac'], d['exchange_checksum'], u['ms_per_step'], u['value'], u['sweep1']['frac']))"
  done
done 2>&1 | tee $OUT/inflight_ab3.txt
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --sustained-steps 0 --u8-images 0"
cd /tmp
rm -rf $OUT/prof_stats
MSFM_IN_FLIGHT=3 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -- $BENCH > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
cd $ROOT
DB=$(ls -t $(find $OUT/prof_stats -name '*.db') | head -1)
python tools/step_timeline.py "$DB" 4 > $OUT/step_timeline_3sets.txt 2>&1; tail -1 $OUT/step_timeline_3sets.txt | cut -c1-300
find $OUT/prof_stats -type f -size +8M -delete
