"""Exact-integer reference of the matcher for INTEGER-valued descriptors (raw SIFT / the u8 configs).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  What it pins: for descriptors whose components are
integers in [0, 255] every partial sum of S(q,t) = sum_c (a_c - b_c)^2 is an integer <= 128 * 255^2 =
8 323 200 < 2^24, hence exact in fp32 under ANY accumulation order, fused or not.  A
conforming cv::hal::normL2Sqr_ build therefore produces exactly the integer computed here in int64, and
sqrtf of an exactly represented fp32 value is correctly rounded on every IEEE implementation -- so for this
input class the distances and kNN indices below are what EVERY OpenCV build must return, independent of the
restated SIMD reduction tree in msfm_oracle.c / np_oracle.py.  The tests assert
    C oracle (all orders) == NumPy-fp32 oracle == this == HIP path
on the u8 fixtures, which pins the integer half of the kNN claim without OpenCV.

Follows (reference file:line):
  knnMatch(k=2) selection rule, sqrt placement      src/Feature/FeatureUtils.cpp:146-149 + SURVEY App. C
  ComputeMatches (strict fp32 ratio test)           src/Feature/FeatureUtils.cpp:141-157
  ComputeCrossMatches / CrossCheck (operator[] quirk) src/Feature/FeatureUtils.cpp:160-174, 281-310
  FilterMatchesByDistance (double compare)          src/Feature/FeatureUtils.cpp:208-218
"""
import numpy as np

F32 = np.float32
FLT_MAX = np.finfo(F32).max


def _as_int(desc):
    d = np.asarray(desc)
    di = d.astype(np.int64)
    if not np.array_equal(di, d) or di.min(initial=0) < 0 or di.max(initial=0) > 255:
        raise ValueError("int_oracle needs integer-valued descriptors in [0, 255]")
    if d.ndim != 2 or (d.shape[0] and d.shape[1] != 128):
        raise ValueError("descriptors must be n x 128")
    return di


def s_matrix(A, B, rows=None):
    """S[q][t] = |a_q - b_t|^2 as int64, all arithmetic in int64 (no floating point anywhere)."""
    A, B = _as_int(A), _as_int(B)
    if rows is not None:
        A = A[rows]
    na = (A * A).sum(1)
    nb = (B * B).sum(1)
    S = na[:, None] + nb[None, :] - 2 * (A @ B.T)   # NumPy's integer matmul: exact int64 products and sums
    assert S.min(initial=0) >= 0 and S.max(initial=0) < (1 << 24)
    return S


def knn2(A, B, rows=None, block=512, tie="lowest"):
    """-> idx0, d0, idx1, d1 like BFMatcher(NORM_L2).knnMatch(A[rows], B, k=2): per query the two smallest
    sqrtf(S) under (distance ascending, train index ascending); -1 / FLT_MAX where fewer neighbours exist.
    tie="highest" flips the tie rule (equal distances: the HIGHER train index first) -- not what batchDistance does
    (SURVEY App. C restates its rule from memory); it exists so that the tests can show which results do NOT depend on it."""
    assert tie in ("lowest", "highest")
    Ai, Bi = _as_int(A), _as_int(B)
    if rows is not None:
        Ai = Ai[np.asarray(rows)]
    n1, n2 = Ai.shape[0], Bi.shape[0]
    idx0 = np.full(n1, -1, np.int32)
    idx1 = np.full(n1, -1, np.int32)
    d0 = np.full(n1, FLT_MAX, F32)
    d1 = np.full(n1, FLT_MAX, F32)
    if n2 == 0:
        return idx0, d0, idx1, d1
    for s in range(0, n1, block):
        S = s_matrix(Ai[s:s + block], Bi)
        D = np.sqrt(S.astype(F32))      # S < 2^24: the conversion is exact; IEEE sqrt is correctly rounded
        # batchDistance compares the distance BIT PATTERNS (non-negative floats: same order) and keeps the lower
        # train index on ties: a stable sort on D is that rule
        if tie == "lowest":
            order = np.argsort(D.view(np.int32), axis=1, kind="stable")[:, :2]
        else:
            order = n2 - 1 - np.argsort(D.view(np.int32)[:, ::-1], axis=1, kind="stable")[:, :2]
        r = np.arange(D.shape[0])
        idx0[s:s + block] = order[:, 0]
        d0[s:s + block] = D[r, order[:, 0]]
        if n2 >= 2:
            idx1[s:s + block] = order[:, 1]
            d1[s:s + block] = D[r, order[:, 1]]
    return idx0, d0, idx1, d1


def compute_matches(A, B, ratio=0.8, tie="lowest"):
    """FeatureUtils::ComputeMatches: keep (q, idx0) iff d0 < fl32(ratio * d1), strict; train < 2 rows -> none."""
    n1, n2 = len(A), len(B)
    if n1 == 0 or n2 < 2:
        z = np.zeros(0, np.int32)
        return z, z.copy(), np.zeros(0, F32)
    i0, d0, _, d1 = knn2(A, B, tie=tie)
    thr = (F32(ratio) * d1).astype(F32)       # one fp32 multiply, single rounding
    keep = d0 < thr
    q = np.nonzero(keep)[0].astype(np.int32)
    return q, i0[keep].astype(np.int32), d0[keep]


def match_pair(A, B, ratio=0.8, cross_check=True, max_distance=0.7, tie="lowest"):
    """ComputeCrossMatches / ComputeMatches + FilterMatchesByDistance, as MatchImagePairs chains them."""
    if cross_check and (len(A) < 2 or len(B) < 2):
        # one direction has a train set of < 2 rows: undefined in the reference (FeatureUtils.cpp:152 indexes the 2nd
        # neighbour unconditionally); build-defined as "no matches" (include/msfm_match.h, msfm_oracle.c)
        z = np.zeros(0, np.int32)
        return z, z.copy(), np.zeros(0, F32)
    q, t, d = compute_matches(A, B, ratio, tie)
    if cross_check:
        rq, rt, _ = compute_matches(B, A, ratio, tie)
        # vis[reverse query] = reverse train; a missing key reads as 0 (unordered_map::operator[], FeatureUtils.cpp:302)
        vis = {int(a): int(b) for a, b in zip(rq, rt)}
        keep = np.array([vis.get(int(tt), 0) == int(qq) for qq, tt in zip(q, t)], bool) if len(q) else np.zeros(0, bool)
        q, t, d = q[keep], t[keep], d[keep]
    keep = ~(d.astype(np.float64) > float(max_distance))
    return q[keep], t[keep], d[keep]
