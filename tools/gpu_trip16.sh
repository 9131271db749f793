#!/bin/bash
# scratch sets in flight x pipeline depth with the direct route-Q thresholds (two matrix kernels per sub-batch instead of three)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for round in 1 2; do
  for cfg in "2 4" "3 4" "3 6" "3 8" "2 6" "1 1"; do
    set -- $cfg
    MSFM_IN_FLIGHT=$1 MSFM_PIPELINE=$2 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('in flight $1 pipeline $2 round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step (frac %.3f), checksum %s | u8 job %.2f ms per step %.3e/s sweep-1 frac %.3f' % (d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], r['frac'], d['exchange_checksum'], u['ms_per_step'], u['value'], u['sweep1']['frac']))"
  done
done 2>&1 | tee $OUT/inflight_ab2.txt
