#!/bin/bash
# Is the integer sweep paced by the delivery of its B tiles?  Timing experiment: a build whose DMA groups always re-read the item's
# FIRST tile (L2-resident after the first touch; results are wrong -- the call then retries on the brute-force route, so the sweep's
# duration is read from a kernel trace -- the instruction stream is the same) against the real build.  Second experiment: the same
# with the per-tile workgroup barrier removed (races: wrong results) -- what the synchronisation of the sixteen waves costs.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_EXPERIMENT_SAME_TILE -shared -o /tmp/libmsfm_sametile.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_EXPERIMENT_NO_BARRIER -shared -o /tmp/libmsfm_nobarrier.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep " error"
cat > /tmp/exp.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from monocularsfm_amd import _lib, synth
imgs, pairs, _ = synth.job("south-building", 64)
ctx = _lib.Context(0)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
for _ in range(3):
    try:
        ctx.match_pairs(pairs)
    except Exception as e:
        print("   (call failed: %s)" % str(e)[:80])
PY
cd /tmp
for lib in $ROOT/monocularsfm_amd/csrc/libmsfm_match.so /tmp/libmsfm_sametile.so /tmp/libmsfm_nobarrier.so; do
  rm -rf /tmp/exp_prof
  MSFM_PIPELINE=1 MSFM_LIBRARY=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/exp_prof -- python /tmp/exp.py > /tmp/exp.log 2>&1
  DB=$(ls -t $(find /tmp/exp_prof -name '*.db') | head -1)
  echo "== $(basename $lib): durations (ms) of the sweep_i8_kernel<1> launches of 3 calls on the 64-image job (2016 pairs, 5.1e10 descriptor pairs)"
  python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
print([round((e - s) / 1e6, 3) for n, s, e in rows if "sweep_i8_kernel<1>" in n])
PY
done 2>&1 | tee $OUT/dma_experiment.txt
