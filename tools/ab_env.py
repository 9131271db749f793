"""A/B of a library switch that is read at msfm_create: two contexts (env var = 1 / 0) resident on one GPU, the bench job
timed in alternation.  Usage: python tools/ab_env.py MSFM_ITEM_PANELS [u8]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

var = sys.argv[1]
u8 = len(sys.argv) > 2 and sys.argv[2] == "u8"
imgs, pairs, _ = synth.job("synthetic-u8", 48, 8192, seed=1329) if u8 else synth.job("south-building", 128)
kw = {"max_distance": 1e9} if u8 else {}
ctxs = {}
for f in ("1", "0"):
    os.environ[var] = f
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctxs[f] = ctx
res = {"1": [], "0": []}
ref = None
for rnd in range(10):
    for f in ("1", "0"):
        offs, qt, d = ctxs[f].match_pairs(pairs, fetch="view", **kw)
        p = ctxs[f].profile()
        if rnd >= 2:
            res[f].append((p["approx_kernel_ms"], p["sweep2_ms"], p["total_device_ms"]))
        cur = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
        if ref is None:
            ref = cur
        assert all(np.array_equal(x, y) for x, y in zip(ref, cur)), "results differ"
for f in ("1", "0"):
    a = np.array(res[f])
    print("%s=%s sweep1 min %.3f med %.3f ms | sweep2 med %.3f | device span med %.3f ms" % (
        var, f, a[:, 0].min(), np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2])), flush=True)
