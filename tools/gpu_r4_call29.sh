#!/bin/bash
# Round 4, call 29: wall clock of the driver's default bench command
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
T0=$(date +%s.%N)
python bench.py > $OUT/r4_bench_default.json 2> $OUT/r4_bench_default.err; echo "bench rc=$?"
T1=$(date +%s.%N)
python - <<PY
import json
print("wall seconds of 'python bench.py':", round($T1 - $T0, 1))
d = json.loads(open('$OUT/r4_bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['solo']['frac'], d['strong_u8']['seconds_per_step'], d['cpu_baseline']['value'])
PY
