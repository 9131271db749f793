#!/bin/bash
# Round 4, call 25: why the transposed reverse items of sweep 2 gain nothing -- per-wave cycle sums of the loop segments and of the item-level
# segments (library built with -DMSFM_SWEEP_PROBE), plain plan vs MSFM_S2_TRANSPOSE=1, bench job with the pipeline off
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_SWEEP_PROBE -shared -o /tmp/libmsfm_probe.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
for T in 0 1; do
MSFM_S2_TRANSPOSE=$T MSFM_PIPELINE=1 MSFM_DEBUG_TIMING=1 MSFM_LIBRARY=/tmp/libmsfm_probe.so timeout 300 python - 128 > $OUT/r4_s2t_probe_$T.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, _ = synth.job("south-building", int(sys.argv[1]) if len(sys.argv) > 1 else 128)
ctx = _lib.Context(0)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
for _ in range(2):
    ctx.match_pairs(pairs); p = ctx.profile(); print("sweep1 %.3f ms sweep2 %.3f ms" % (p["approx_kernel_ms"], p["sweep2_ms"]), flush=True)
PY
echo "== MSFM_S2_TRANSPOSE=$T"; grep "sweep 3 probe" -A1 $OUT/r4_s2t_probe_$T.txt | tail -6 | cut -c1-330; grep "^sweep1" $OUT/r4_s2t_probe_$T.txt | tail -1; grep "msfm plan" $OUT/r4_s2t_probe_$T.txt | tail -1 | cut -c1-250
done
