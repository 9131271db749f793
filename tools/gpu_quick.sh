#!/bin/bash
# Quick loop: prefilter parity tests, the bench line, sweep-1 HBM traffic (FETCH_SIZE pass) -- a few minutes
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_i8.py tests/test_gpu_jobs.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1]); r=d["roofline"]
print("value %.4g ms/step %.2f sweep1 %.2f (frac %.3f) sweep2 %.2f step/sweep1 %.3f" % (d["value"], d["ms_per_step"], r["sweep1_ms_per_step"], r["frac"], r["sweep2"]["ms_per_step"], r["step_over_sweep1"]))
u=d["strong_u8"]; print("u8 value %.4g ms/step %.1f sweep1 frac %.3f" % (u["value"], u["ms_per_step"], u["sweep1"]["frac"]))
PY
export TMPDIR=/tmp; cd /tmp; rm -rf $OUT/pmc_fetch
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --u8-images 0 > $OUT/pmc_fetch.log 2>&1
cd $ROOT; python tools/pmc_summary.py $OUT/pmc_traffic_quick.json "sweep_kernel<1>,sweep_kernel<3>" $OUT/pmc_fetch | grep -A2 FETCH | grep KB
find $OUT/pmc_fetch -type f -size +8M -delete
