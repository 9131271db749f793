"""The exact-integer (int64) reference pins the integer half of the kNN claim without OpenCV: for descriptors
with integer components in [0, 255] every conforming build of cv::hal::normL2Sqr_ returns the same bits
(SURVEY.md App. C), so   C oracle (every accumulation order) == NumPy-fp32 oracle == int64 reference
must hold -- here on seeded u8 sets with planted ties and on the committed u8 fixtures.  The HIP path is held
to the same reference in tests/test_gpu_jobs.py.  (/root/reference/src/Feature/FeatureUtils.cpp:141-174, 281-310)"""
import glob
import os

import numpy as np
import pytest

from monocularsfm_amd import synth
from oracle import int_oracle as io, np_oracle as no

HERE = os.path.dirname(__file__)


def b(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def tie_set(seed, n1, n2):
    u = synth.u8_images(2, [n1, n2], seed=seed, dup_frac=0.15)
    u[1][7] = u[1][3]
    u[1][min(90, n2 - 1)] = u[1][3]
    u[0][10] = u[1][3]          # distance 0 three times: index ties
    u[0][0] = u[1][40]          # queryIdx 0 has a twin: the operator[] quirk path of CrossCheck
    return u


@pytest.mark.parametrize("shape", [(300, 280), (131, 5), (64, 700), (2, 2), (1, 3), (3, 1)])
def test_c_and_numpy_oracles_equal_the_integer_reference(oracle, shape):
    n1, n2 = shape
    u = tie_set(1000 + n1 + 7 * n2, max(n1, 12), max(n2, 100))
    A, B = u[0][:n1], u[1][:n2]
    ref = io.knn2(A, B)
    for order in (0, 1, 2):
        got = oracle.knn2(A, B, order, 4)
        for x, y in zip(ref, got):
            assert np.array_equal(b(x), b(y)), order
    got = no.knn2(A, B, 0)
    for x, y in zip(ref, got):
        assert np.array_equal(b(x), b(y))
    for cc in (True, False):
        for ratio, md in ((0.8, 1e9), (0.95, 400.0), (1.0, 1e9)):
            r = io.match_pair(A, B, ratio, cc, md)
            for order in (0, 1):
                g = oracle.match_pair(A, B, ratio, cc, md, order)
                for x, y in zip(r, g):
                    assert np.array_equal(b(x), b(y)), (cc, ratio, order)


def row_with_norm(s):
    """A descriptor row (integers 0..255) whose squared norm is exactly s: 255s, then a four-square remainder."""
    v = np.zeros(128, np.float32)
    k = s // 65025
    v[:k] = 255
    r = s - k * 65025
    for a in range(int(r ** 0.5), -1, -1):
        for b_ in range(int((r - a * a) ** 0.5), -1, -1):
            rest = r - a * a - b_ * b_
            c = int(rest ** 0.5)
            while c >= 0:
                d = rest - c * c
                e = int(round(d ** 0.5))
                if e * e == d and e <= 255:
                    v[k:k + 4] = (a, b_, c, e)
                    return v
                c -= 1
    raise AssertionError("no decomposition")


def test_sqrt_collisions_of_large_integer_s(oracle):
    """S >= 2^22: consecutive integers share a sqrtf; the LOWER train index must win although its S is larger
    (batchDistance compares sqrt'ed distances) -- identical in the C oracle and the integer reference."""
    s = np.arange(6_000_000, 6_000_400, dtype=np.int64)
    d = np.sqrt(s.astype(np.float32)).view(np.int32)
    k = int(np.nonzero(np.diff(d) == 0)[0][0])
    s_big, s_small = int(s[k + 1]), int(s[k])          # equal sqrtf, s_big > s_small
    A = np.zeros((2, 128), np.float32)                   # query 0 = origin: S(q, t) = |b_t|^2
    B = np.stack([row_with_norm(s_big), row_with_norm(s_small), row_with_norm(s_big + 4000), row_with_norm(s_small)])
    assert [int(x) for x in io.s_matrix(A[:1], B)[0]] == [s_big, s_small, s_big + 4000, s_small]
    ref = io.knn2(A, B)
    assert ref[0][0] == 0 and ref[2][0] == 1 and ref[1][0] == ref[3][0]   # index 0 wins the sqrt-space tie
    for order in (0, 1, 2):
        got = oracle.knn2(A, B, order)
        for x, y in zip(ref, got):
            assert np.array_equal(b(x), b(y))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "u8_*.npz"))),
                         ids=lambda p: os.path.basename(p))
def test_integer_reference_reproduces_the_u8_fixtures(path):
    g = np.load(path)
    A, B = g["desc1"], g["desc2"]
    f, r = io.knn2(A, B), io.knn2(B, A)
    for o in (0, 1):
        if "o%d_fwd_idx0" % o not in g:
            continue
        assert np.array_equal(f[0], g["o%d_fwd_idx0" % o]) and np.array_equal(b(f[1]), b(g["o%d_fwd_d0" % o]))
        assert np.array_equal(b(f[3]), b(g["o%d_fwd_d1" % o]))
        assert np.array_equal(r[0], g["o%d_rev_idx0" % o]) and np.array_equal(b(r[1]), b(g["o%d_rev_d0" % o]))
        for cc in (1, 0):
            q, t, d = io.match_pair(A, B, 0.8, bool(cc), float(g["max_distance"]))
            assert np.array_equal(q, g["o%d_cc%d_q" % (o, cc)]) and np.array_equal(t, g["o%d_cc%d_t" % (o, cc)])
            assert np.array_equal(b(d), b(g["o%d_cc%d_d" % (o, cc)]))
    if "int_fwd_s0" in g:   # fixtures written from the integer reference carry the integer S of both neighbours
        S = io.s_matrix(A, B)
        assert np.array_equal(np.sort(S, 1)[:, 0], g["int_fwd_s0"]) and np.array_equal(np.sort(S, 1)[:, 1], g["int_fwd_s1"])


def test_match_lists_do_not_depend_on_the_tie_rule():
    """ratio <= 1: a row with d0 == d1 fails `d0 < ratio * d1` whichever tied index comes first, so flipping the tie rule
    (highest train index first) changes knnMatch's idx0 on the planted ties but not one entry of the match lists
    (FeatureUtils.cpp:146-156; the header states it next to msfm_fetch_order_certificate)."""
    g = np.load(os.path.join(HERE, "golden", "u8_ties_150x161.npz"))
    A, B = g["desc1"], g["desc2"]
    lo, hi = io.knn2(A, B), io.knn2(A, B, tie="highest")
    assert np.array_equal(b(lo[1]), b(hi[1])) and np.array_equal(b(lo[3]), b(hi[3]))     # the distance VALUES never depend on it
    flipped = np.nonzero(lo[0] != hi[0])[0]
    assert len(flipped) >= 1 and (lo[1][flipped] == lo[3][flipped]).all()                  # idx0 differs exactly on d0 == d1 rows
    for ratio in (0.8, 0.95, 1.0):
        for cc in (True, False):
            ref = io.match_pair(A, B, ratio, cc, 1e9)
            alt = io.match_pair(A, B, ratio, cc, 1e9, tie="highest")
            assert len(ref[0]) > 0 and all(np.array_equal(b(x), b(y)) for x, y in zip(ref, alt)), (ratio, cc)
    # ... and with ratio > 1 it DOES matter (which is why the library runs the tie fix-up there)
    ref = io.match_pair(A, B, 1.5, False, 1e9)
    alt = io.match_pair(A, B, 1.5, False, 1e9, tie="highest")
    assert not np.array_equal(ref[1], alt[1])


def test_rejects_non_integer_input():
    with pytest.raises(ValueError):
        io.knn2(np.full((2, 128), 0.5, np.float32), np.zeros((2, 128), np.float32))
