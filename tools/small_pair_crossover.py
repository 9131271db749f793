"""Where does the MFMA prefilter start to pay?  4096 pairs of n x n descriptors, prefilter on / off."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

ctx = _lib.Context(0)
for n in (50, 100, 128, 200, 256, 384, 512, 768, 1024):
    N = 92  # 4186 pairs
    imgs = synth.rootsift_images(N, n, seed=n, n_proto=4 * n)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    pairs = np.array([(i, j) for i in range(N) for j in range(i)], np.int32)
    out = {}
    for pf in (True, False):
        ctx.set_prefilter(pf)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            offs, _, _ = ctx.match_pairs(pairs, fetch=False)
            best = min(best, time.perf_counter() - t0)
        out[pf] = (best, int(offs[-1]))
    ctx.set_prefilter(True)
    assert out[True][1] == out[False][1]
    print("n = %4d: prefilter %.3f ms, brute %.3f ms  (%.2f us / %.2f us per pair)  -> %s" % (
        n, out[True][0] * 1e3, out[False][0] * 1e3, out[True][0] / len(pairs) * 1e6, out[False][0] / len(pairs) * 1e6,
        "prefilter" if out[True][0] < out[False][0] else "brute"), flush=True)
