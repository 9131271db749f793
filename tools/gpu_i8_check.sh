#!/bin/bash
# One GPU call for the integer-core path: its parity tests, the quick three-route table, loop-segment probe on a byte job
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_i8.py tests/test_gpu_jobs.py -m gpu -x -q > $OUT/pytest_i8.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_i8.log
timeout 300 python tools/configs_table.py --quick > $OUT/configs_quick.txt 2>&1; echo "table rc=$?"; cat $OUT/configs_quick.txt | cut -c1-260
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_SWEEP_PROBE -shared -o /tmp/libmsfm_probe.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
MSFM_DEBUG_TIMING=1 MSFM_LIBRARY=/tmp/libmsfm_probe.so timeout 300 python - > $OUT/probe_i8.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, _ = synth.job("synthetic-u8", 32, 8192, seed=1329)
ctx = _lib.Context(0)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
for mode in (1, 2):
    ctx.set_prefilter(mode)
    for _ in range(2):
        ctx.match_pairs(pairs, max_distance=1e9); p = ctx.profile(); print("mode %d sweep1 %.3f ms sweep2 %.3f ms" % (mode, p["approx_kernel_ms"], p["sweep2_ms"]), flush=True)
PY
grep -v "msfm host" $OUT/probe_i8.txt | tail -24
