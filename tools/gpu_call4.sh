#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_clock tools/ubench_clock.hip 2>&1 | grep error; /tmp/ubench_clock > $OUT/ubench_clock.txt 2>&1; cat $OUT/ubench_clock.txt
# clocks / power while the bench loop runs
(timeout 120 python bench.py --steps 400 --warmup 2 --no-cpu-baseline --u8-images 0 > $OUT/bench_long.json 2>$OUT/bench_long.err) &
sleep 45
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -6; sleep 3; done > $OUT/smi.txt 2>&1
wait
cat $OUT/smi.txt | head -30
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_long.json')); r=d['roofline']
print(d['ms_per_step'], r['sweep1_ms_per_step'], r['frac'], r['sweep2']['ms_per_step'])
PY
