import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
ctx = _lib.Context(0)
for (n1, n2) in ((64, 64), (64, 65), (65, 64), (128, 64), (64, 128), (33, 64), (64, 300), (2, 300), (512, 512), (513, 64)):
    imgs = synth.rootsift_images(2, [n1, n2], seed=n1 + 3 * n2, n_proto=max(n1, n2) * 2)
    ctx.upload_image(0, imgs[0]); ctx.upload_image(1, imgs[1])
    ctx.knn2_pair(0, 1); p = ctx.profile()
    ctx.match_pair(0, 1, 0.8, True, float("inf")); q = ctx.profile()
    print((n1, n2), "knn: pf %d fb %d cand %d | match: pf %d fb %d cand %d compact %d" % (
        p["prefilter_pairs"], p["fallback_pairs"], p["candidates"], q["prefilter_pairs"], q["fallback_pairs"], q["candidates"], q["compacted_pairs"]), flush=True)
