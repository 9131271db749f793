"""A/B of two builds of the library on the bench job: contexts of both resident on one GPU, timed in alternation.
Usage: python tools/ab_libs.py <base.so> [u8]      (the other build is the in-tree one)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

u8 = len(sys.argv) > 2 and sys.argv[2] == "u8"
imgs, pairs, _ = synth.job("synthetic-u8", 48, 8192, seed=1329) if u8 else synth.job("south-building", 128)
kw = {"max_distance": 1e9} if u8 else {}
tree = _lib.LIB_PATH
ctxs = {}
for name, path in (("tree", tree), ("base", sys.argv[1])):
    _lib._lib = None
    _lib.LIB_PATH = path
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctxs[name] = ctx
res = {k: [] for k in ctxs}
ref = None
for rnd in range(10):
    for name, ctx in ctxs.items():
        offs, qt, d = ctx.match_pairs(pairs, fetch="view", **kw)
        p = ctx.profile()
        if rnd >= 2:
            res[name].append((p["approx_kernel_ms"], p["sweep2_ms"], p["total_device_ms"]))
        cur = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
        if ref is None:
            ref = cur
        assert all(np.array_equal(x, y) for x, y in zip(ref, cur)), "results differ"
for name in ctxs:
    a = np.array(res[name])
    print("%s: sweep1 min %.3f med %.3f ms | sweep2 med %.3f | device span med %.3f ms" % (
        name, a[:, 0].min(), np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2])), flush=True)
