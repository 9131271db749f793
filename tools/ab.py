"""A/B/C... on ONE box: one context per variant, all resident, the job timed in alternation (round-robin, so that clock drift and the
box's thermal state hit every variant alike).  A variant is a BUILD of the library and / or a set of SWITCHES read at msfm_create:

    python tools/ab.py [--u8] [--p1] [--images N] [--rounds R] [--nocheck] VARIANT [VARIANT ...]

    VARIANT = [name=][path/to/libmsfm_match.so][@ENV=VAL[,ENV2=VAL2...]]
        tree                        the in-tree build, default switches
        r05=/tmp/w/libmsfm_match.so another build (git worktree add /tmp/w <commit> && make -C /tmp/w/monocularsfm_amd/csrc)
        @MSFM_PIPELINE=3            the in-tree build with a switch
        q=tree@MSFM_Q8=0,MSFM_PIPELINE=1

Per variant: sub-batches, sweep-1 / sweep-2 event times, the call's device span and wall clock (median / min), sweep-1 rate; the variants'
results must be identical (--nocheck: timing experiments with wrong results).  --p1: pipeline off (one sub-batch, every kernel alone).
(Replaces ab_env.py, ab_envs.py, ab_libs.py and ab_multi.py of rounds 2-5 -- the same loop four times.)"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monocularsfm_amd import _lib, synth  # noqa: E402


def parse(spec):
    name, _, rest = spec.partition("=") if ("=" in spec.split("@", 1)[0]) else ("", "", spec)
    lib, _, envs = rest.partition("@")
    env = dict(x.split("=", 1) for x in envs.split(",") if x)
    label = name or (spec if spec else "defaults")
    return label, (lib if lib and lib != "tree" else None), env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--u8", action="store_true")
    ap.add_argument("--p1", action="store_true", help="pipeline off: one sub-batch, every kernel alone")
    ap.add_argument("--images", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--nocheck", action="store_true", help="timing experiments with wrong results: do not compare the variants' lists")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    imgs, pairs, name = synth.job("synthetic-u8", args.images or 48, 8192, seed=1329) if args.u8 else synth.job("south-building", args.images or 128)
    kw = {"max_distance": 1e9} if args.u8 else {}
    tree, all_exports = _lib.LIB_PATH, list(_lib.EXPORTS)
    ctxs = {}
    for spec in args.variants:
        label, lib, env = parse(spec)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        _lib._lib = None
        _lib.LIB_PATH = lib or tree
        probe = ctypes.CDLL(_lib.LIB_PATH)
        _lib.EXPORTS = [e for e in all_exports if hasattr(probe, e)]   # (an older build may lack the newest entry points)
        ctx = _lib.Context(0)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        if args.p1:
            ctx.set_pipeline(1)
        ctxs[label] = ctx
    res = {k: [] for k in ctxs}
    ref = None
    for rnd in range(args.rounds):
        for nm, ctx in ctxs.items():
            t0 = time.perf_counter()
            offs, qt, d = ctx.match_pairs(pairs, fetch="view", **kw)
            wall = (time.perf_counter() - t0) * 1e3
            p = ctx.profile()
            if rnd >= 2:
                res[nm].append((p["approx_kernel_ms"], p["sweep2_ms"], p["total_device_ms"], wall, p["sub_batches"],
                                256e-12 * p["prefilter_descriptor_pairs"] / max(1e-9, p["approx_kernel_ms"])))
            cur = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
            if ref is None:
                ref = cur
            assert args.nocheck or all(np.array_equal(x, y) for x, y in zip(ref, cur)), "results differ: " + nm
    print("# %s%s, %d rounds" % (name, " (pipeline off)" if args.p1 else "", args.rounds - 2))
    for nm in ctxs:
        a = np.array(res[nm])
        print("%-44s sub-batches %d | sweep1 med %.3f ms (%.3f POP/s over the pairs it swept) | sweep2 med %.3f | device span min %.3f med %.3f ms | wall med %.3f ms" % (
            nm, int(a[0, 4]), np.median(a[:, 0]), np.median(a[:, 5]), np.median(a[:, 1]), a[:, 2].min(), np.median(a[:, 2]), np.median(a[:, 3])), flush=True)


if __name__ == "__main__":
    main()
