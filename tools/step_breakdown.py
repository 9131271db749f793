"""Where one bench step goes: wall time of msfm_match_pairs (device + host orchestration), of the fetch,
and the kernel-side counters.  Usage: python tools/step_breakdown.py [n_images]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(1234)
counts = rng.integers(4600, 5401, N)
imgs = synth.rootsift_images(N, counts.tolist(), seed=1234, n_proto=20000, sigma=0.05)
ctx = _lib.Context(0)
for i, im in enumerate(imgs):
    ctx.upload_image(i, im)
pairs = np.array([(i, j) for i in range(N) for j in range(i)], np.int32)
for rep in range(4):
    t0 = time.perf_counter()
    offs, _, _ = ctx.match_pairs(pairs, fetch=False)
    t1 = time.perf_counter()
    offs, qt, d = ctx.match_pairs(pairs, fetch=True)
    t2 = time.perf_counter()
    offs, qt, d = ctx.match_pairs(pairs, fetch="view")
    t3 = time.perf_counter()
    p = ctx.profile()
    print("rep %d: match_pairs(no fetch) %.1f ms | with fetch %.1f ms | view %.1f ms | device span %.1f ms | sweep1 %.1f ms sweep2 %.1f ms exact-path %.1f ms | matches %d" % (
        rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, p["total_device_ms"], p["approx_kernel_ms"], p["sweep2_ms"], p["dist_kernel_ms"], offs[-1]))
