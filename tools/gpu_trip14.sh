#!/bin/bash
# route Q with direct thresholds (adaptive twin level): tests, fuzz in the three modes, bench A/B against the refinement sweep
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
MSFM_Q8=2 timeout 400 python tools/fuzz_routes.py 301 500 > $OUT/fuzz_q8_auto.txt 2>&1; echo "fuzz auto rc=$?"; tail -2 $OUT/fuzz_q8_auto.txt
MSFM_Q8=2 MSFM_Q8_DIRECT=2 timeout 400 python tools/fuzz_routes.py 302 500 > $OUT/fuzz_q8_direct.txt 2>&1; echo "fuzz direct rc=$?"; tail -2 $OUT/fuzz_q8_direct.txt
MSFM_Q8=2 MSFM_Q8_DIRECT=0 timeout 400 python tools/fuzz_routes.py 303 300 > $OUT/fuzz_q8_refine.txt 2>&1; echo "fuzz refine rc=$?"; tail -2 $OUT/fuzz_q8_refine.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --sustained-steps 0 --u8-images 0 > $OUT/bench_direct_$i.json 2> $OUT/bench_direct.err; echo "bench direct rc=$?"
MSFM_Q8_DIRECT=0 timeout 300 python bench.py --no-cpu-baseline --sustained-steps 0 --u8-images 0 > $OUT/bench_refine_$i.json 2> $OUT/bench_refine.err; echo "bench refine rc=$?"
done
python - <<'PY'
import json
for n in ("direct_1", "refine_1", "direct_2", "refine_2"):
    try:
        d = json.loads(open("gpurun_out/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), d.get("kernel_ms_per_step"), d.get("candidates_per_step"))
    except Exception as e:
        print(n, "failed", e)
PY
