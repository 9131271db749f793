"""A/B/C... of several builds of the library on one box: one context per build, all resident, the job timed in alternation.
Usage: python tools/ab_multi.py [--u8] [--p1] [--images N] [--rounds R] name=path.so [name=path.so ...]   ("tree" = the in-tree build)
Per build: sweep-1 / sweep-2 event times, the call's device span and wall clock (median / min); results must be identical."""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--u8", action="store_true")
ap.add_argument("--p1", action="store_true", help="pipeline off: one sub-batch, every kernel alone")
ap.add_argument("--images", type=int, default=None)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--nocheck", action="store_true", help="timing experiments with wrong results: do not compare the builds' lists")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
imgs, pairs, name = synth.job("synthetic-u8", args.images or 48, 8192, seed=1329) if args.u8 else synth.job("south-building", args.images or 128)
kw = {"max_distance": 1e9} if args.u8 else {}
tree = _lib.LIB_PATH
all_exports = list(_lib.EXPORTS)
ctxs = {}
for spec in args.libs:
    nm, _, path = spec.partition("=")
    _lib._lib = None
    _lib.LIB_PATH = tree if (nm == "tree" and not path) else path
    import ctypes
    probe = ctypes.CDLL(_lib.LIB_PATH)
    _lib.EXPORTS = [e for e in all_exports if hasattr(probe, e)]   # (an older build may lack the newest entry points)
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    if args.p1:
        ctx.set_pipeline(1)
    ctxs[nm] = ctx
res = {k: [] for k in ctxs}
ref = None
for rnd in range(args.rounds):
    for nm, ctx in ctxs.items():
        t0 = time.perf_counter()
        offs, qt, d = ctx.match_pairs(pairs, fetch="view", **kw)
        wall = (time.perf_counter() - t0) * 1e3
        p = ctx.profile()
        if rnd >= 2:
            res[nm].append((p["approx_kernel_ms"], p["sweep2_ms"], p["total_device_ms"], wall, 256e-12 * p["prefilter_descriptor_pairs"] / max(1e-9, p["approx_kernel_ms"])))
        cur = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
        if ref is None:
            ref = cur
        assert args.nocheck or all(np.array_equal(x, y) for x, y in zip(ref, cur)), "results differ: " + nm
print("# %s%s, %d rounds" % (name, " (pipeline off)" if args.p1 else "", args.rounds - 2))
for nm in ctxs:
    a = np.array(res[nm])
    print("%-12s sweep1 med %.3f ms (%.3f POP/s over the pairs it swept) | sweep2 med %.3f | device span min %.3f med %.3f ms | wall med %.3f ms" % (
        nm, np.median(a[:, 0]), np.median(a[:, 4]), np.median(a[:, 1]), a[:, 2].min(), np.median(a[:, 2]), np.median(a[:, 3])), flush=True)
