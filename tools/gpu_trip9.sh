#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 31 300 > $OUT/fuzz_a.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_a.txt
for round in 1 2; do
  for p in 1 4; do
    MSFM_PIPELINE=$p timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('pipeline $p round $round: %.2f ms per step, %.3e desc-pairs/s, sweep 1 %.2f ms per step (frac %.3f), sweep 1b %.2f ms, sweep 2 %.2f ms, checksum %s | u8 job %.2f ms per step %.3e/s sweep-1 frac %.3f' % (d['ms_per_step'], d['value'], r['sweep1_ms_per_step'], r['frac'], r['route_q']['sweep1b_ms_per_step'], r['sweep2']['ms_per_step'], d['exchange_checksum'], u['ms_per_step'], u['value'], u['sweep1']['frac']))"
  done
done 2>&1 | tee $OUT/rowdigits_ab.txt
timeout 900 python tools/config4_full.py --int-oracle-pairs 1 > $OUT/config4_full.json 2> $OUT/config4_full.err; echo "config4 rc=$?"; head -c 700 $OUT/config4_full.json
