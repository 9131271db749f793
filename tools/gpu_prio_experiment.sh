#!/bin/bash
# wave priority in the integer sweep: matrix halves at priority 1 (shipped) / no priorities / epilogue halves at priority 1
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for m in 1 2; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_I8_PRIO_MODE=$m -shared -o /tmp/libmsfm_prio$m.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
done
for round in 1 2 3; do
for lib in $ROOT/monocularsfm_amd/csrc/libmsfm_match.so /tmp/libmsfm_prio1.so /tmp/libmsfm_prio2.so; do
MSFM_PIPELINE=1 MSFM_LIBRARY=$lib python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, _ = synth.job("south-building", 96)
ctx = _lib.Context(0)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
t = []
for _ in range(5):
    ctx.match_pairs(pairs); t.append(ctx.profile()["approx_kernel_ms"])
print(os.path.basename(os.environ["MSFM_LIBRARY"]), "sweep 1 ms per launch (96 images):", ["%.2f" % x for x in t[1:]], "offsets[-1]", int(ctx.match_pairs(pairs)[0][-1]))
PY
done; done 2>&1 | tee $OUT/prio_experiment.txt
