"""Order-invariance certificate (msfm_fetch_order_certificate, msfm_profile.order_sensitive_rows).

The reference delegates S(q,t) to cv::BFMatcher::knnMatch (/root/reference/src/Feature/FeatureUtils.cpp:146-156), an
unpinned OpenCV whose fp32 accumulation order depends on the build.  The certificate marks every row whose decisions
(first neighbour, d0 < ratio * d1, d0 <= max_distance) are within the worst-case reassociation bound of flipping;
zero marked rows => the stored (queryIdx, trainIdx) lists cannot depend on the order.  Checked here:
  * the device count == a NumPy restatement of the predicate on the knnMatch-level arrays, on both routes;
  * certificate == 0  =>  the two named orders (SSE 4x4 and AVX2+FMA) give the same lists (config-2-shaped data);
  * planted near-ties at the ratio boundary and at the distance cut MUST fire;
  * byte images are exact under every order: never marked."""
import numpy as np
import pytest

from monocularsfm_amd import _lib, synth

pytestmark = pytest.mark.gpu
F32 = np.float32
EPS = 1.0e-5
FLT_MAX = np.finfo(F32).max


def predicate(i0, d0, d1, ratio, max_distance, forward):
    """csrc/msfm_kernels.hip.h order_sensitive(), vectorised in float64 like the device code"""
    i0, d0, d1 = np.asarray(i0), np.asarray(d0, F32), np.asarray(d1, F32)
    has = (i0 >= 0) & (d1 < FLT_MAX)
    lo0, hi0 = d0.astype(np.float64) * (1 - EPS), d0.astype(np.float64) * (1 + EPS)
    lo1, hi1 = d1.astype(np.float64) * (1 - EPS), d1.astype(np.float64) * (1 + EPS)
    cut = (lo0 > max_distance) if forward else np.zeros(len(d0), bool)
    passes = d0 < (F32(ratio) * d1).astype(F32)
    r = float(F32(ratio))
    cert_pass = (hi0 < r * lo1) & (hi0 < lo1)
    if forward:
        cert_pass &= (hi0 <= max_distance) | (lo0 > max_distance)
    cert_fail = lo0 >= r * hi1
    certified = np.where(passes, cert_pass, cert_fail)
    return has & ~cut & ~certified


def expected_count(ctx, i, j, ratio=0.8, cross_check=True, max_distance=0.7):
    (fi, fd0, fd1), (ri, rd0, rd1) = ctx.knn2_pair(int(i), int(j))
    n = int(predicate(fi, fd0, fd1, ratio, max_distance, True).sum())
    if cross_check:
        n += int(predicate(ri, rd0, rd1, ratio, max_distance, False).sum())
    return n


def test_certificate_is_clean_on_rootsift_data_and_the_orders_agree(gpu_ctx):
    imgs, pairs, _ = synth.job("south-building", 24, seed=1234)
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    offs, qt, d = gpu_ctx.match_pairs(pairs)
    prof = gpu_ctx.profile()
    cert = gpu_ctx.order_certificate(len(pairs))
    assert offs[-1] > 10000
    assert prof["order_sensitive_rows"] == int(cert.sum())
    # the device count is the predicate's count (sampled pairs; the knnMatch-level call keeps every row alive)
    for p in (0, 7, 100, len(pairs) - 1):
        assert cert[p] == expected_count(gpu_ctx, *pairs[p]), p
    # brute-force route: same certificate
    gpu_ctx.set_prefilter(False)
    try:
        gpu_ctx.match_pairs(pairs[:40])
        assert np.array_equal(gpu_ctx.order_certificate(40), cert[:40])
    finally:
        gpu_ctx.set_prefilter(True)
    # certificate == 0 on a pair  =>  the other named orders (AVX2+FMA, AVX-512+FMA) return the same index list for it
    assert (cert == 0).sum() > 0.9 * len(pairs)       # unit-norm RootSIFT data: margins are orders of magnitude above 1e-5
    for other_order in (_lib.ORDER_AVX2_FMA, _lib.ORDER_AVX512_FMA):
        with _lib.Context(0, order=other_order) as other:
            for i, im in enumerate(imgs):
                other.upload_image(i, im)
            offs2, qt2, d2 = other.match_pairs(pairs)
            assert np.array_equal(other.order_certificate(len(pairs)) == 0, cert == 0)    # (the certificate itself is stable)
        differing_distances = 0
        for p in range(len(pairs)):
            a, b_ = qt[offs[p]:offs[p + 1]], qt2[offs2[p]:offs2[p + 1]]
            if cert[p] == 0:
                assert np.array_equal(a, b_), (other_order, p)
                differing_distances += int((d[offs[p]:offs[p + 1]].view(np.int32) != d2[offs2[p]:offs2[p + 1]].view(np.int32)).sum())
        assert differing_distances > 0                 # ... although the orders do differ in the low bits of the distances


def planted():
    """Image A: 3 query rows; image B: rows whose distances to the queries sit ON a decision boundary.
       q0: d0 = 0.4, d1 = 0.5 (1 + 1e-6)     -> ratio test within 1e-6 of flipping          MUST fire
       q1: d0 = 0.7 (1 - 2e-6), d1 = 1.2     -> passes the ratio test, straddles the cut     MUST fire
       q2: d0 = 0.3, d1 = 0.9                -> clear pass, clear of the cut                 certified
    (a shared offset of 3.0 in a private dimension keeps the three groups 4.2 apart)"""
    def unit(k, v):
        x = np.zeros(128, F32)
        x[k] = v
        return x
    A = np.stack([unit(100, 3.0), unit(101, 3.0), unit(102, 3.0)])
    B = np.stack([
        unit(0, 0.4) + unit(100, 3.0), unit(3, 0.5 * (1 + 1e-6)) + unit(100, 3.0),
        unit(1, 0.7 * (1 - 2e-6)) + unit(101, 3.0), unit(4, 1.2) + unit(101, 3.0),
        unit(2, 0.3) + unit(102, 3.0), unit(5, 0.9) + unit(102, 3.0),
    ])
    return A.astype(F32), B.astype(F32)


@pytest.mark.parametrize("prefilter", [True, False])
def test_planted_near_ties_fire(gpu_ctx, prefilter):
    A, B = planted()
    gpu_ctx.upload_image(0, A)
    gpu_ctx.upload_image(1, B)
    gpu_ctx.set_prefilter(prefilter)
    try:
        for cross in (False, True):
            gpu_ctx.match_pairs([(0, 1)], cross_check=cross)
            cert = gpu_ctx.order_certificate(1)
            assert cert[0] == expected_count(gpu_ctx, 0, 1, cross_check=cross)
            (fi, fd0, fd1), _ = gpu_ctx.knn2_pair(0, 1)
            s = predicate(fi, fd0, fd1, 0.8, 0.7, True)
            assert list(s) == [True, True, False]
            assert cert[0] >= 2
        # far from every boundary with a looser ratio and cut: clean
        gpu_ctx.match_pairs([(0, 1)], ratio=0.95, cross_check=False, max_distance=5.0)
        assert gpu_ctx.order_certificate(1)[0] == 0 and gpu_ctx.profile()["order_sensitive_rows"] == 0
    finally:
        gpu_ctx.set_prefilter(True)


def test_byte_images_are_exact_under_every_order(gpu_ctx):
    u = synth.u8_images(3, [700, 640, 900], seed=5, dup_frac=0.3, as_float=False)
    for i, x in enumerate(u):
        gpu_ctx.upload_image(i, x)
    pairs = synth.all_pairs(3)
    offs, _, _ = gpu_ctx.match_pairs(pairs, max_distance=1e9)
    assert offs[-1] > 100 and gpu_ctx.profile()["order_sensitive_rows"] == 0
    assert not gpu_ctx.order_certificate(len(pairs)).any()
    # the same values uploaded as floats are recognised as a byte store (every value is checked at upload): still exact
    for i, x in enumerate(u):
        gpu_ctx.upload_image(i, x.astype(F32))
    gpu_ctx.match_pairs(pairs, max_distance=1e9)
    assert gpu_ctx.profile()["order_sensitive_rows"] == 0 and not gpu_ctx.order_certificate(len(pairs)).any()
    # ... half-integers are not: the predicate runs (and may or may not fire)
    for i, x in enumerate(u):
        gpu_ctx.upload_image(i, x.astype(F32) + F32(0.5))
    gpu_ctx.match_pairs(pairs, max_distance=1e9)
    cert = gpu_ctx.order_certificate(len(pairs))
    for p in range(len(pairs)):
        assert cert[p] == expected_count(gpu_ctx, *pairs[p], max_distance=1e9)


def test_match_lists_do_not_depend_on_the_tie_rule(gpu_ctx):
    """The other half of SURVEY App. C: batchDistance's tie rule (lower train index among equal distances) is restated from
    memory like the accumulation order.  On the planted-tie fixture the device lists equal the integer reference with the rule
    FLIPPED (highest index first) for every ratio <= 1 -- a job with 0 order-sensitive rows is independent of both
    (/root/reference/src/Feature/FeatureUtils.cpp:146-156; include/msfm_match.h next to msfm_fetch_order_certificate)."""
    import os
    from oracle import int_oracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "u8_ties_150x161.npz"))
    A, B = g["desc1"], g["desc2"]
    lo, hi = int_oracle.knn2(A, B), int_oracle.knn2(A, B, tie="highest")
    assert (lo[0] != hi[0]).any()                                    # the fixture does contain first-place ties
    for as_bytes in (True, False):                                   # byte upload / the same values in the CV_32F table
        gpu_ctx.upload_image(0, A.astype(np.uint8) if as_bytes else A.astype(F32))
        gpu_ctx.upload_image(1, B.astype(np.uint8) if as_bytes else B.astype(F32))
        for prefilter in (1, 2, 0):                                  # integer cores / fp16 cores + exact re-check / brute force
            gpu_ctx.set_prefilter(prefilter)
            try:
                for ratio in (0.8, 1.0):
                    for cc in (True, False):
                        q, t, d = gpu_ctx.match_pair(0, 1, ratio, cc, 1e9)
                        rq, rt, rd = int_oracle.match_pair(A, B, ratio, cc, 1e9, tie="highest")
                        assert len(rq) > 0 and np.array_equal(q, rq) and np.array_equal(t, rt), (as_bytes, prefilter, ratio, cc)
                        assert np.array_equal(d.view(np.int32), np.asarray(rd, F32).view(np.int32))
                assert gpu_ctx.profile()["order_sensitive_rows"] == 0       # byte values: exact under every order
            finally:
                gpu_ctx.set_prefilter(True)
    # the knnMatch-level API does see the rule, and implements the LOWER index
    (fi, fd0, fd1), _ = gpu_ctx.knn2_pair(0, 1)
    assert np.array_equal(fi, lo[0]) and not np.array_equal(fi, hi[0])
