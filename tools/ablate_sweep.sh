#!/bin/bash
# Builds the library with each MSFM_ABL variant into /tmp and times sweep 1 / sweep 2 on the 32 x 5000 job
# (results of the ablated builds are wrong by construction; only the kernel times matter).
set -u
ROOT=$(pwd)
for v in 0 1 2 3 4 5 6 7 8; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -DMSFM_ABL=$v -shared \
      -o /tmp/libmsfm_abl$v.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep -E "error" 
done
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs = synth.rootsift_images(32, 5000, seed=11)
pairs = np.array([(i, j) for i in range(32) for j in range(i)], np.int32)
names = {0: "full", 1: "no epilogue", 2: "no MFMA", 3: "no LDS B reads", 4: "no barrier/DMA", 5: "no epi + no LDS", 6: "MFMA only", 7: "barrier every 2nd tile", 8: "no DMA wait"}
for v in range(9):
    _lib._lib = None
    _lib.LIB_PATH = "/tmp/libmsfm_abl%d.so" % v
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs): ctx.upload_image(i, im)
    for rep in range(3):
        try:
            ctx.match_pairs(pairs, fetch=False)
        except Exception as e:
            print("  (error: %s)" % e); break
        p = ctx.profile()
    print("ABL %d %-16s sweep1 %.2f ms  sweep2 %.2f ms (compacted %d, fallback %d)" % (v, names[v], p["approx_kernel_ms"], p["sweep2_ms"], p["compacted_pairs"], p["fallback_pairs"]), flush=True)
    ctx.close()
PY
