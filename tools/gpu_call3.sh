#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py --cpu-budget 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1200 $OUT/bench.json; tail -3 $OUT/bench.err
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_SWEEP_PROBE -shared -o /tmp/libmsfm_probe.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep error
MSFM_LIBRARY=/tmp/libmsfm_probe.so python - > $OUT/probe.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs = synth.rootsift_images(32, 5000, seed=11)
pairs = synth.all_pairs(32)
ctx = _lib.Context(0)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
for _ in range(3):
    ctx.match_pairs(pairs); p = ctx.profile(); print("sweep1 %.3f ms sweep2 %.3f ms" % (p["approx_kernel_ms"], p["sweep2_ms"]), flush=True)
PY
tail -9 $OUT/probe.txt
