#!/bin/bash
# pipeline depth on route Q (alternated twice)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for round in 1 2; do
  for p in 2 5 6 8 10 12 16; do
    MSFM_PIPELINE=$p timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 0 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('route Q pipeline $p round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step (frac %.3f), sweep 1b %.2f ms, sweep 2 %.2f ms, sub-batches %d, checksum %s' % (d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], r['frac'], r['route_q']['sweep1b_ms_per_step'], r['sweep2']['ms_per_step'], d['sub_batches_per_step'], d['exchange_checksum']))"
  done
done 2>&1 | tee $OUT/pipeline_ab_q8.txt
