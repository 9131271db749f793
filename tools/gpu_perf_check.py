"""Quick on-GPU A/B harness: parity on two shapes + kernel throughput on a 32-image all-pairs job."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocularsfm_amd import _lib, synth
from oracle import c_oracle as co

def eq(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and bool((a.view(np.int32) == b.view(np.int32)).all())

orders = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
ctx = _lib.Context(0)
ok_all = True
for order in orders:
    ctx.set_accum_order(order)
    for (n1, n2, seed) in [(600, 500, 1), (129, 257, 2), (1500, 1400, 6)]:
        imgs = synth.rootsift_images(2, [n1, n2], seed=seed, n_proto=max(n1, n2) * 2)
        ctx.upload_image(0, imgs[0]); ctx.upload_image(1, imgs[1])
        (fi, fd0, fd1), (ri, rd0, rd1) = ctx.knn2_pair(0, 1)
        oi0, od0, oi1, od1 = co.knn2(imgs[0], imgs[1], order, 8)
        pi0, pd0, pi1, pd1 = co.knn2(imgs[1], imgs[0], order, 8)
        k_ok = eq(fi, oi0) and eq(fd0, od0) and eq(fd1, od1) and eq(ri, pi0) and eq(rd0, pd0) and eq(rd1, pd1)
        print("order", order, (n1, n2), "knn2", k_ok)
        ok_all &= k_ok
    N, n = 32, 5000
    imgs = synth.rootsift_images(N, n, seed=11)
    for i, im in enumerate(imgs): ctx.upload_image(i, im)
    pairs = np.array([(i, j) for i in range(N) for j in range(i)], np.int32)
    ref = None
    for pf in (False, True):
        ctx.set_prefilter(pf)
        for rep in range(3):
            t0 = time.time(); offs, qt, d = ctx.match_pairs(pairs); dt = time.time() - t0
            prof = ctx.profile()
        if not pf:
            ref = (offs.copy(), qt.copy(), d.copy())
            best = prof["descriptor_pairs"] / prof["dist_kernel_ms"] * 1e3
            print("order %d brute : kernel %.1f ms  desc-pairs/s (kernel) %.4e  = %.1f T lane-ops/s  wall %.3f s" % (
                order, prof["dist_kernel_ms"], best, best * (384 if order == 0 else 256) / 1e12, dt))
        else:
            same = eq(offs, ref[0]) and eq(qt, ref[1]) and eq(d, ref[2])
            ok_all &= same
            print("order %d prefilter: sweep1 %.2f ms (%.0f TFLOP/s f16)  sweep2 %.2f ms (%d compacted pairs, %.3f of the dense work)  wall %.3f s  (%.3e desc-pairs/s)  cand/row %.2f  fallback %d  identical %s" % (
                order, prof["approx_kernel_ms"],
                256 * prof["prefilter_descriptor_pairs"] / max(prof["approx_kernel_ms"], 1e-9) * 1e3 / 1e12,
                prof["sweep2_ms"], prof["compacted_pairs"], prof["sweep2_descriptor_pairs"] / max(1, prof["prefilter_descriptor_pairs"]), dt,
                prof["descriptor_pairs"] / dt, prof["candidates"] / (2 * N * (N - 1) / 2 * n), prof["fallback_pairs"], same))
print("ALL OK" if ok_all else "FAILURES")
