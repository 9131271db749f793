"""One library build, one job, a few calls -- for a rocprofv3 kernel trace around it (tools/gpu_r4_call27.sh lists every launch of sweep 1
per build: timing experiments with wrong results flood their candidate lists and re-run smaller sweeps, which an event SUM hides).
Usage: MSFM_LIBRARY=path.so python tools/s1_launches.py [--u8] [--images N] [--calls K]"""
import argparse
import sys

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--u8", action="store_true")
ap.add_argument("--images", type=int, default=40)
ap.add_argument("--calls", type=int, default=4)
args = ap.parse_args()
imgs, pairs, name = synth.job("synthetic-u8", args.images, 8192, seed=1329) if args.u8 else synth.job("south-building", args.images)
kw = {"max_distance": 1e9} if args.u8 else {}
ctx = _lib.Context(0)
for i, im in enumerate(imgs):
    ctx.upload_image(i, im)
ctx.set_pipeline(1)
for _ in range(args.calls):
    ctx.match_pairs(pairs, fetch="view", **kw)
