"""The bench job with ONE image that has a value outside the unit interval (no byte twin): what a single outlier costs.  Round 3 / early round 4: every
pair of every sub-batch that holds one of its pairs fell back to the fp16 first sweep; now only the outlier's pairs do.
Usage: python tools/mixed_store_ab.py [images]   (prints per-call device span and the route counters, homogeneous vs one outlier)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
imgs, pairs, name = synth.job("south-building", n)
ctx = _lib.Context(0)
outlier = imgs[5].copy()
outlier[0, int(np.argmin(outlier[0]))] = -1e-3       # one value outside [0, 1]: no byte twin for the image
for tag in ("homogeneous", "image 5: one value < 0 (no twin)"):
    for i, im in enumerate(imgs):
        ctx.upload_image(i, outlier if (i == 5 and tag != "homogeneous") else im)
    t = []
    for _ in range(6):
        ctx.match_pairs(pairs, fetch="view")
        p = ctx.profile()
        t.append((p["total_device_ms"], p["approx_kernel_ms"]))
    t = np.array(t[2:])
    print("%-32s device span med %.2f ms | sweep 1 (both launches) %.2f ms | route Q launches %d, mixed sub-batches %d, demoted pairs %d, sub-batches %d" % (
        tag, np.median(t[:, 0]), np.median(t[:, 1]), p["sweep1_q8_launches"], p["mixed_route_sub_batches"], p["demoted_pairs"], p["sub_batches"]), flush=True)
