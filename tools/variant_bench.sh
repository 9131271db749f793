#!/bin/bash
# VB_WORKLOAD=u8: byte images (integer-core sweeps).
# usage: tools/variant_bench.sh "<name>=<extra hipcc flags>" ...   -> sweep times of each build on the 32 x 5000 job
set -u
ROOT=$(pwd)
names=()
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include $flags -shared \
      -o /tmp/libmsfm_var_$name.so $ROOT/monocularsfm_amd/csrc/msfm_match.hip 2>&1 | grep -E "error"
  names+=("$name")
done
python - "${names[@]}" <<'PY'
import sys, ctypes, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
import os
U8 = os.environ.get("VB_WORKLOAD", "") == "u8"     # byte store: the integer-core sweeps
imgs = synth.u8_images(32, 5000, seed=11, as_float=False) if U8 else synth.rootsift_images(32, 5000, seed=11)
kw = {"max_distance": 1e9} if U8 else {}
pairs = np.array([(i, j) for i in range(32) for j in range(i)], np.int32)
names = sys.argv[1:]
ctxs = {}
for name in names:          # one context per variant, all resident at once: the variants are timed in alternation
    _lib._lib = None
    _lib.LIB_PATH = "/tmp/libmsfm_var_%s.so" % name
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs): ctx.upload_image(i, im)
    ctxs[name] = ctx
res = {n: ([], []) for n in names}
ref = None
same = {}
for rnd in range(30):
    for name in names:
        ctx = ctxs[name]
        offs, qt, d = ctx.match_pairs(pairs, **kw)
        p = ctx.profile()
        if rnd >= 5:
            res[name][0].append(p["approx_kernel_ms"]); res[name][1].append(p["sweep2_ms"])
        if rnd == 0:
            if ref is None: ref = (offs.copy(), qt.copy(), d.copy())
            same[name] = np.array_equal(offs, ref[0]) and np.array_equal(qt, ref[1]) and np.array_equal(d.view(np.int32), ref[2].view(np.int32))
for name in names:
    s1, s2 = res[name]
    print("%-14s sweep1 min %.3f med %.3f ms | sweep2 min %.3f med %.3f ms | same as first: %s" % (
        name, min(s1), sorted(s1)[len(s1) // 2], min(s2), sorted(s2)[len(s2) // 2], same[name]), flush=True)
PY
