"""Opportunistic pin of the oracle against a REAL OpenCV: runs wherever `cv2` is importable and is
skipped otherwise (this image and the GPU boxes have no OpenCV, which is why DESIGN.md says "parity unpinned").

* integer-valued descriptors: every partial sum is exact in fp32, so cv2.BFMatcher(NORM_L2).knnMatch(k=2) must
  agree with the oracle bit for bit under ANY build;
* float descriptors: the low bits depend on the build's SIMD tree -- one of the oracle's named accumulation orders
  must reproduce it (which one is printed)."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from monocularsfm_amd import synth  # noqa: E402


def cv_knn2(a, b):
    m = cv2.BFMatcher(cv2.NORM_L2).knnMatch(np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32), k=2)
    idx0 = np.array([r[0].trainIdx for r in m], np.int32)
    d0 = np.array([r[0].distance for r in m], np.float32)
    d1 = np.array([r[1].distance for r in m], np.float32)
    return idx0, d0, d1


def test_integer_descriptors_match_opencv_bit_for_bit(oracle):
    u = synth.u8_images(2, [400, 350], seed=5, as_float=True)
    ci, cd0, cd1 = cv_knn2(u[0], u[1])
    for order in (0, 1):
        oi, od0, _, od1 = oracle.knn2(u[0], u[1], order, 4)
        assert np.array_equal(oi, ci)
        assert np.array_equal(od0.view(np.int32), cd0.view(np.int32)) and np.array_equal(od1.view(np.int32), cd1.view(np.int32))


def test_float_descriptors_match_one_named_order(oracle):
    imgs = synth.rootsift_images(2, [500, 450], seed=6, n_proto=900)
    ci, cd0, cd1 = cv_knn2(imgs[0], imgs[1])
    hits = []
    for order, name in ((0, "MSFM_ORDER_SSE4X4"), (1, "MSFM_ORDER_AVX2_FMA")):
        oi, od0, _, od1 = oracle.knn2(imgs[0], imgs[1], order, 4)
        if np.array_equal(oi, ci) and np.array_equal(od0.view(np.int32), cd0.view(np.int32)) and np.array_equal(od1.view(np.int32), cd1.view(np.int32)):
            hits.append(name)
    print("OpenCV", cv2.__version__, "reproduced by:", hits)
    assert hits, "this OpenCV build sums in an order the oracle does not name (add it to oracle/msfm_oracle.c and the kernels)"


def test_ratio_crosscheck_distance_pipeline_against_opencv(oracle):
    """ComputeCrossMatches + FilterMatchesByDistance rebuilt from cv2's knnMatch, on integer data (order-free)."""
    u = synth.u8_images(2, [300, 280], seed=7, as_float=True)
    a, b = (u[0] / 512.0).astype(np.float32), (u[1] / 512.0).astype(np.float32)   # powers of two keep the sums exact

    def one_way(x, y):
        i0, d0, d1 = cv_knn2(x, y)
        keep = d0 < np.float32(0.8) * d1
        return np.nonzero(keep)[0].astype(np.int32), i0[keep], d0[keep]

    q12, t12, d12 = one_way(a, b)
    q21, t21, _ = one_way(b, a)
    vis = dict(zip(q21.tolist(), t21.tolist()))
    keep = np.array([vis.get(int(t), 0) == int(q) for q, t in zip(q12, t12)], bool)   # operator[] quirk: missing key reads 0
    q, t, d = q12[keep], t12[keep], d12[keep]
    keep = ~(d.astype(np.float64) > 0.7)
    oq, ot, od = oracle.match_pair(a, b, 0.8, True, 0.7, nthreads=4)
    assert np.array_equal(oq, q[keep]) and np.array_equal(ot, t[keep]) and np.array_equal(od.view(np.int32), d[keep].view(np.int32))
