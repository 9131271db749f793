#!/bin/bash
# 4-byte packed column partials of the integer sweeps: parity (tests + fuzz), then A/B against the float2 build
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 901 900 > $OUT/fuzz_a.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_a.txt
timeout 400 python tools/fuzz_routes.py 902 900 > $OUT/fuzz_b.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_b.txt
for round in 1 2 3; do
  for lib in csrc/libmsfm_match_prev.so csrc/libmsfm_match.so; do
    MSFM_LIBRARY=$ROOT/monocularsfm_amd/$lib timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('$lib round $round: %.2f ms per step | solo sweep 1 %.2f ms (frac %.3f), unpipelined step %.2f | cand/row %.4f | u8 job %.2f ms per step %.3e/s solo frac %.3f | checksum %s' % (d['ms_per_step'], r['solo']['avg_launch_ms'], r['solo']['frac'], r['solo']['ms_per_step_unpipelined'], r['candidates_per_row'], u['ms_per_step'], u['value'], u['sweep1']['solo_frac'], d['exchange_checksum']))"
  done
done 2>&1 | tee $OUT/cp_pack_ab.txt
