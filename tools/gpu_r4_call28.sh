#!/bin/bash
# Round 4, call 28: wall clock of the driver's default bench command; a long route fuzz on the final build (new seeds)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
/usr/bin/time -v python bench.py > $OUT/r4_bench_default.json 2> $OUT/r4_bench_default.err; echo "bench rc=$?"; grep "Elapsed (wall clock)\|Maximum resident" $OUT/r4_bench_default.err; python -c "
import json; d=json.loads(open('$OUT/r4_bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['solo']['frac'], d['strong_u8']['seconds_per_step'], d['cpu_baseline']['value'])"
timeout 900 python tools/fuzz_routes.py 1201 2500 > $OUT/r4_fuzz_long_default.txt 2>&1; echo "rc=$?"; tail -1 $OUT/r4_fuzz_long_default.txt
MSFM_Q8=2 timeout 900 python tools/fuzz_routes.py 1301 2500 > $OUT/r4_fuzz_long_q8.txt 2>&1; echo "rc=$?"; tail -1 $OUT/r4_fuzz_long_q8.txt
