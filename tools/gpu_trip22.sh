#!/bin/bash
# fewer VALU instructions per tile in the integer sweep: parity + A/B (solo sweep times) against the previous build
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_i8.py tests/test_gpu_q8.py tests/test_gpu_jobs.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for round in 1 2 3; do
  for lib in csrc/libmsfm_match_prev.so csrc/libmsfm_match.so; do
    [ -f $ROOT/monocularsfm_amd/$lib ] || continue
    MSFM_LIBRARY=$ROOT/monocularsfm_amd/$lib timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --u8-images 192 --u8-steps 3 --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; u = d['strong_u8']
print('$lib round $round: %.2f ms per step | solo sweep 1 %.2f ms (frac %.3f), unpipelined step %.2f | u8 job %.2f ms per step %.3e/s solo frac %.3f' % (d['ms_per_step'], r['solo']['avg_launch_ms'], r['solo']['frac'], r['solo']['ms_per_step_unpipelined'], u['ms_per_step'], u['value'], u['sweep1']['solo_frac']))"
  done
done 2>&1 | tee $OUT/valu_ab.txt
