"""A/B of two builds of the library on the bench job: contexts of both resident on one GPU, timed in alternation.
Usage: python tools/ab_libs.py <base.so> [u8] [p1]      (the other build is the in-tree one; p1: pipeline off, one sub-batch)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

u8 = "u8" in sys.argv[2:]
p1 = "p1" in sys.argv[2:]
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
    if p1:
        ctx.set_pipeline(1)
    ctxs[name] = ctx
res = {k: [] for k in ctxs}
ref = None
for rnd in range(10):
    for name, ctx in ctxs.items():
        t0 = time.perf_counter()
        offs, qt, d = ctx.match_pairs(pairs, fetch="view", **kw)
        wall = (time.perf_counter() - t0) * 1e3
        p = ctx.profile()
        if rnd >= 2:
            res[name].append((p["approx_kernel_ms"], p["sweep2_ms"], p["total_device_ms"], wall))
        cur = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
        if ref is None:
            ref = cur
        assert all(np.array_equal(x, y) for x, y in zip(ref, cur)), "results differ"
for name in ctxs:
    a = np.array(res[name])
    print("%s%s: sweep1 min %.3f med %.3f ms | sweep2 med %.3f | device span med %.3f ms | wall per call min %.3f med %.3f ms" % (
        name, " (pipeline off)" if p1 else "", a[:, 0].min(), np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]),
        a[:, 3].min(), np.median(a[:, 3])), flush=True)
