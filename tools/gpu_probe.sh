#!/bin/bash
# Diagnostic: per-wave cycle sums of the sweep loop segments and of the item-level segments (library built with -DMSFM_SWEEP_PROBE)
# on the bench job (route Q: the integer sweep on the twins + the compacted fp16 sweep 2), one sub-batch per call
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_SWEEP_PROBE -shared -o /tmp/libmsfm_probe.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
MSFM_PIPELINE=1 MSFM_DEBUG_TIMING=1 MSFM_LIBRARY=/tmp/libmsfm_probe.so python - ${1:-128} > $OUT/probe.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, _ = synth.job("south-building", int(sys.argv[1]) if len(sys.argv) > 1 else 128)
ctx = _lib.Context(0)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
for _ in range(2):
    ctx.match_pairs(pairs); p = ctx.profile(); print("sweep1 %.3f ms sweep2 %.3f ms" % (p["approx_kernel_ms"], p["sweep2_ms"]), flush=True)
PY
grep -v "msfm host\|msfm plan" $OUT/probe.txt | tail -24 | cut -c1-330
