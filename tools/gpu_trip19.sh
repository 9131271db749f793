#!/bin/bash
# one barrier per tile in the integer sweep: parity tests, then A/B against the two-barrier build (libmsfm_match_twobarrier.so)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_i8.py tests/test_gpu_q8.py tests/test_gpu_jobs.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 300 python tools/fuzz_routes.py 401 300 > $OUT/fuzz_1b.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_1b.txt
for round in 1 2 3; do
  for lib in csrc/libmsfm_match_twobarrier.so csrc/libmsfm_match.so; do
    MSFM_LIBRARY=$ROOT/monocularsfm_amd/$lib MSFM_PIPELINE=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('$lib round $round (one sub-batch): %.2f ms per step, sweep 1 %.2f ms per step (frac %.3f), sweep 2 %.2f, checksum %s | u8 job %.2f ms per step %.3e/s sweep-1 frac %.3f' % (d['ms_per_step'], r['sweep1_ms_per_step'], r['frac'], r['sweep2']['ms_per_step'], d['exchange_checksum'], u['ms_per_step'], u['value'], u['sweep1']['frac']))"
  done
done 2>&1 | tee $OUT/onebarrier_ab.txt
