"""A/B of library switches that are read at msfm_create: one context per environment setting, all resident, the job timed in alternation.
Usage: python tools/ab_envs.py [--u8] [--images N] [--rounds R] "NAME=VAL,NAME2=VAL2" "NAME=VAL" ...      ("" = the defaults)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--u8", action="store_true")
ap.add_argument("--images", type=int, default=None)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("envs", nargs="+")
args = ap.parse_args()
imgs, pairs, name = synth.job("synthetic-u8", args.images or 48, 8192, seed=1329) if args.u8 else synth.job("south-building", args.images or 128)
kw = {"max_distance": 1e9} if args.u8 else {}
ctxs = {}
for spec in args.envs:
    kv = dict(x.split("=", 1) for x in spec.split(",") if x)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    ctx = _lib.Context(0)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctxs[spec or "defaults"] = ctx
res = {k: [] for k in ctxs}
ref = None
for rnd in range(args.rounds):
    for nm, ctx in ctxs.items():
        t0 = time.perf_counter()
        offs, qt, d = ctx.match_pairs(pairs, fetch="view", **kw)
        wall = (time.perf_counter() - t0) * 1e3
        p = ctx.profile()
        if rnd >= 2:
            res[nm].append((p["approx_kernel_ms"], p["total_device_ms"], wall, p["sub_batches"], p["sweep2_ms"]))
        cur = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
        if ref is None:
            ref = cur
        assert all(np.array_equal(x, y) for x, y in zip(ref, cur)), "results differ: " + nm
print("# %s, %d rounds" % (name, args.rounds - 2))
for nm in ctxs:
    a = np.array(res[nm])
    print("%-44s sub-batches %d | sweep1 med %.3f ms | sweep2 med %.3f ms | device span min %.3f med %.3f ms | wall med %.3f ms" % (
        nm, int(a[0, 3]), np.median(a[:, 0]), np.median(a[:, 4]), a[:, 1].min(), np.median(a[:, 1]), np.median(a[:, 2])), flush=True)
