"""SURVEY 8f-3 at scale: the pre-emptive test (100 top-scale descriptors per image, cross-matched, >= 4 matches keeps the
pair) of ALL pairs of a large image set as batched launches: N images x 100 rows -> N (N - 1) / 2 pairs through
msfm_match_pairs (sub-batches of 16384 pairs).  Prints the rate and what config 5's 8.4 M pairs would take.
Usage: python tools/preemptive_scale.py [n_images]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imgs = synth.rootsift_images(N, 100, seed=5, n_proto=3000)
ctx = _lib.Context(0)
for i, im in enumerate(imgs):
    ctx.upload_image(i, im)
pairs = synth.all_pairs(N)
for rep in range(3):
    t0 = time.perf_counter()
    offs, _, _ = ctx.match_pairs(pairs, ratio=0.8, cross_check=True, max_distance=np.inf, fetch=False)
    dt = time.perf_counter() - t0
    p = ctx.profile()
    keep = int((np.diff(offs) >= 4).sum())
    print("%d images x 100 rows, %d pairs: %.1f ms = %.3f us per pair (%d sub-batches, device %.1f ms); %d pairs kept; "
          "8 386 560 pairs (config 5) at this rate: %.2f s" % (N, len(pairs), dt * 1e3, dt / len(pairs) * 1e6, p["sub_batches"],
                                                                p["total_device_ms"], keep, dt / len(pairs) * 8386560), flush=True)
