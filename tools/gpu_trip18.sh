#!/bin/bash
# taper of the parts x sets in flight x parts
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for round in 1 2; do
  for cfg in "2 4 1.0" "3 4 1.0" "2 4 0.5" "3 4 0.5" "3 4 0.3" "3 5 0.4" "3 6 0.3" "2 5 0.4" "3 3 0.5" "2 3 0.5"; do
    set -- $cfg
    MSFM_IN_FLIGHT=$1 MSFM_PIPELINE=$2 MSFM_PIPELINE_TAPER=$3 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('in flight $1 parts $2 taper $3 round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step, sub-batches %s, checksum %s | u8 job %.2f ms per step %.3e/s' % (d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], d['sub_batches_per_step'], d['exchange_checksum'], u['ms_per_step'], u['value']))"
  done
done 2>&1 | tee $OUT/inflight_ab4.txt
