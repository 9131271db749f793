"""First on-GPU sanity run: parity of the HIP path against the CPU oracle + a rough throughput figure."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monocularsfm_amd import _lib, synth
from oracle import c_oracle as co

def eq(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and bool((a.view(np.int32) == b.view(np.int32)).all())

ctx = _lib.Context(0)
print(ctx.device_info())
ok_all = True
for order in (0, 1):
    ctx.set_accum_order(order)
    for (n1, n2, seed) in [(600, 500, 1), (129, 257, 2), (128, 128, 3), (1000, 77, 4), (2, 2, 5), (5000, 4800, 6), (300, 300, 7)]:
        imgs = synth.rootsift_images(2, [n1, n2], seed=seed, n_proto=max(n1, n2) * 2)
        ctx.upload_image(0, imgs[0]); ctx.upload_image(1, imgs[1])
        (fi, fd0, fd1), (ri, rd0, rd1) = ctx.knn2_pair(0, 1)
        oi0, od0, oi1, od1 = co.knn2(imgs[0], imgs[1], order, 8)
        pi0, pd0, pi1, pd1 = co.knn2(imgs[1], imgs[0], order, 8)
        k_ok = eq(fi, oi0) and eq(fd0, od0) and eq(fd1, od1) and eq(ri, pi0) and eq(rd0, pd0) and eq(rd1, pd1)
        q, t, d = ctx.match_pair(0, 1)
        oq, ot, od = co.match_pair(imgs[0], imgs[1], order=order, nthreads=8)
        m_ok = eq(q, oq) and eq(t, ot) and eq(d, od)
        print("order", order, (n1, n2), "knn2", k_ok, "match", m_ok, "n_matches", len(q))
        if not k_ok:
            bad = np.nonzero(fi != oi0)[0][:5]
            print("  fwd idx mismatches", bad, fi[bad], oi0[bad], fd0[bad], od0[bad])
            bad = np.nonzero(fd0.view(np.int32) != od0.view(np.int32))[0][:5]
            print("  fwd d0 mismatches", bad, fd0[bad], od0[bad])
            bad = np.nonzero(ri != pi0)[0][:5]
            print("  rev idx mismatches", bad, ri[bad], pi0[bad])
        ok_all &= k_ok and m_ok
# u8 data + ties
ctx.set_accum_order(0)
u = synth.u8_images(2, [700, 650], seed=9)
u[1][5] = u[1][3]; u[1][100] = u[1][3]  # duplicate train rows -> exact ties
ctx.upload_image(0, u[0]); ctx.upload_image(1, u[1].astype(np.uint8))
(fi, fd0, fd1), (ri, rd0, rd1) = ctx.knn2_pair(0, 1)
oi0, od0, oi1, od1 = co.knn2(u[0], u[1], 0, 8)
pi0, pd0, pi1, pd1 = co.knn2(u[1], u[0], 0, 8)
k_ok = eq(fi, oi0) and eq(fd0, od0) and eq(fd1, od1) and eq(ri, pi0) and eq(rd0, pd0) and eq(rd1, pd1)
print("u8 + duplicates knn2", k_ok)
ok_all &= k_ok
# throughput: 32 images x 5000, all pairs
N, n = 32, 5000
imgs = synth.rootsift_images(N, n, seed=11)
for i, im in enumerate(imgs): ctx.upload_image(i, im)
pairs = np.array([(i, j) for i in range(N) for j in range(i)], np.int32)
for rep in range(2):
    t0 = time.time(); offs, qt, d = ctx.match_pairs(pairs); dt = time.time() - t0
    prof = ctx.profile()
    print("pairs", len(pairs), "wall %.3fs" % dt, "kernel %.1f ms" % prof["dist_kernel_ms"], "desc-pairs/s (kernel) %.3e" % (prof["descriptor_pairs"] / prof["dist_kernel_ms"] * 1e3),
          "(wall) %.3e" % (prof["descriptor_pairs"] / dt), "matches", offs[-1])
# spot check 3 pairs of the batch against the oracle
for p in (0, 100, len(pairs) - 1):
    i, j = pairs[p]
    oq, ot, od = co.match_pair(imgs[i], imgs[j], nthreads=8)
    s, e = offs[p], offs[p + 1]
    ok = eq(qt[s:e, 0], oq) and eq(qt[s:e, 1], ot) and eq(d[s:e], od)
    print("batch pair", p, (i, j), "ok", ok, len(oq)); ok_all &= ok
print("ALL OK" if ok_all else "FAILURES")
sys.exit(0 if ok_all else 1)
