"""Known-answer tests pinning the CPU oracle to the behavioural spec.

The reference has no test or golden vector for this path and its arithmetic lives in an absent,
un-pinned OpenCV, so these hand-computed cases (plus the independent NumPy restatement) are what
the oracle is pinned by -- "parity unpinned" in the strict sense; see oracle/msfm_oracle.h.
"""
import numpy as np
import pytest

from oracle import np_oracle as no

F32 = np.float32
FLT_MAX = np.finfo(F32).max


def unit(k, v=1.0):
    d = np.zeros(128, F32)
    d[k] = v
    return d


def bits(a):
    return np.asarray(a, F32).view(np.int32)


# ---- S(a,b) -------------------------------------------------------------------------------

def test_l2sqr_hand_values(oracle):
    a = np.zeros(128, F32)
    b = np.zeros(128, F32)
    a[0], a[1], b[0] = 3, 4, 1
    for order in (0, 1, 2, 3, 100):
        assert oracle.l2sqr(a, b, order) == 2 * 2 + 4 * 4
    # all ones vs zeros: 128 exactly in every order
    for order in (0, 1, 2, 3):
        assert oracle.l2sqr(np.ones(128, F32), np.zeros(128, F32), order) == 128.0


def test_l2sqr_orders_differ_on_floats_and_agree_on_integers(oracle):
    rng = np.random.default_rng(5)
    A = rng.random((40, 128), dtype=F32)
    B = rng.random((40, 128), dtype=F32)
    s = {o: np.array([oracle.l2sqr(a, b, o) for a, b in zip(A, B)], F32) for o in (0, 1, 2)}
    assert (bits(s[0]) != bits(s[1])).any() and (bits(s[0]) != bits(s[2])).any()  # build-dependent low bits
    Ai = rng.integers(0, 256, (40, 128)).astype(F32)
    Bi = rng.integers(0, 256, (40, 128)).astype(F32)
    si = {o: np.array([oracle.l2sqr(a, b, o) for a, b in zip(Ai, Bi)], F32) for o in (0, 1, 2)}
    exact = ((Ai.astype(np.int64) - Bi.astype(np.int64)) ** 2).sum(axis=1)
    for o in (0, 1, 2):
        assert np.array_equal(si[o].astype(np.int64), exact)  # integers < 2^24: exact under any order


def test_sse_intrinsics_equal_plain_c_statement(oracle):
    rng = np.random.default_rng(6)
    A = rng.normal(size=(200, 128)).astype(F32)
    B = rng.normal(size=(200, 128)).astype(F32)
    x = np.array([oracle.l2sqr(a, b, 0) for a, b in zip(A, B)], F32)
    y = np.array([oracle.l2sqr(a, b, 100) for a, b in zip(A, B)], F32)
    assert np.array_equal(bits(x), bits(y))


def test_fused_order_intrinsics_equal_their_plain_c_statements(oracle):
    """Round 6: the AVX2+FMA3 and AVX-512 orders run on real intrinsics where the host has them (the CPU baseline's fast side); orders
    101 / 103 are their plain-C statements.  Same lane partials, same reduction trees: the same bits -- on random rows, on rows of very
    different magnitude (cancellation in the trees), on byte-valued rows -- and through the cache-blocked kNN loop that the pair-parallel
    baseline times."""
    import ctypes as C
    L = oracle.lib()
    L.orc_simd_level.restype = C.c_int
    level = L.orc_simd_level()
    assert level in (0, 1, 3)            # AVX-512F hosts have AVX2 + FMA3
    rng = np.random.default_rng(66)
    A = rng.normal(size=(400, 128)).astype(F32) * rng.choice([1e-3, 1.0, 255.0], size=(400, 1)).astype(F32)
    B = rng.normal(size=(400, 128)).astype(F32) * rng.choice([1e-3, 1.0, 255.0], size=(400, 1)).astype(F32)
    A[:50] = rng.integers(0, 256, (50, 128)).astype(F32)
    B[:50] = rng.integers(0, 256, (50, 128)).astype(F32)
    for order, plain in ((1, 101), (3, 103)):
        x = np.array([oracle.l2sqr(a, b, order) for a, b in zip(A, B)], F32)
        y = np.array([oracle.l2sqr(a, b, plain) for a, b in zip(A, B)], F32)
        assert np.array_equal(bits(x), bits(y)), order
    from oracle import np_oracle
    q, t = np.abs(A[:130]) / 300, np.abs(B[:257]) / 300
    for order in (1, 3):
        got = oracle.knn2_blocked(q, t, order)
        ref = oracle.knn2(q, t, order)
        for g, r in zip(got, ref):
            assert np.array_equal(g.view(np.int32), r.view(np.int32))
    if level & 1:
        # and the NumPy restatement of the AVX2 order (an independent statement) agrees with the intrinsics
        s_np = np_oracle.l2sqr_matrix(q[:20], t[:30], order=1) if hasattr(np_oracle, "l2sqr_matrix") else None
        if s_np is not None:
            s_c = np.array([[oracle.l2sqr(a, b, 1) for b in t[:30]] for a in q[:20]], F32)
            assert np.array_equal(bits(s_np.astype(F32)), bits(s_c))


def test_sse_order_is_not_fused(oracle):
    # t*t is rounded before the add: pick t^2 with low bits that an FMA would keep.
    a = np.zeros(128, F32)
    b = np.zeros(128, F32)
    t = F32(1 + 2.0 ** -12)            # t^2 = 1 + 2^-11 + 2^-24 -> rounds to 1 + 2^-11 in fp32
    a[0], a[16] = t, F32(2.0 ** 12)    # same lane partial (k = 0 and 16)
    got = oracle.l2sqr(a, b, 0)
    unfused = F32(F32(t * t) + F32(2.0 ** 24))
    assert got == unfused


@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_c_oracle_matches_numpy_oracle(oracle, order):
    rng = np.random.default_rng(10 + order)
    A = rng.random((150, 128), dtype=F32)
    B = rng.random((131, 128), dtype=F32)
    S = no.l2sqr_matrix(A[:9], B[:11], order)
    Sc = np.array([[oracle.l2sqr(a, b, order) for b in B[:11]] for a in A[:9]], F32)
    assert np.array_equal(bits(S), bits(Sc))
    r1 = oracle.knn2(A, B, order, nthreads=3)
    r2 = no.knn2(A, B, order)
    for x, y in zip(r1, r2):
        assert np.array_equal(bits(x) if x.dtype == F32 else x, bits(y) if y.dtype == F32 else y)


# ---- knnMatch(k=2) ---------------------------------------------------------------------------

def test_knn2_hand_case(oracle):
    # train rows at distances 5, 3, 4 from the query (3-4-5 triangles in dims 0/1)
    q = np.zeros((1, 128), F32)
    t = np.stack([unit(0, 5), unit(1, 3), unit(2, 4)])
    i0, d0, i1, d1 = oracle.knn2(q, t)
    assert (i0[0], d0[0], i1[0], d1[0]) == (1, 3.0, 2, 4.0)


def test_knn2_tie_goes_to_lowest_train_index(oracle):
    q = np.zeros((1, 128), F32)
    t = np.stack([unit(0, 2), unit(5, 1), unit(9, 1), unit(3, 1)])  # three at distance 1
    i0, d0, i1, d1 = oracle.knn2(q, t)
    assert (i0[0], i1[0], d0[0], d1[0]) == (1, 2, 1.0, 1.0)


def test_knn2_tie_is_decided_in_sqrt_space(oracle):
    # two different S with the same sqrtf: the LOWER index must win although its S is larger
    s_small = F32(1.0)
    s_big = np.nextafter(F32(1.0), F32(2.0))
    assert np.sqrt(s_big, dtype=F32) == np.sqrt(s_small, dtype=F32) == 1.0
    q = np.zeros((1, 128), F32)
    t = np.zeros((3, 128), F32)
    t[0, 0] = 9.0
    t[1, 0] = 1.0
    t[1, 1] = 2.0 ** -11.5                    # S = 1 + ~2^-23 -> rounds to nextafter(1) > 1
    t[2, 0] = 1.0                             # S == 1
    S1 = oracle.l2sqr(q[0], t[1])
    assert S1 > 1.0 and np.sqrt(F32(S1)) == 1.0, "test construction"
    i0, d0, i1, d1 = oracle.knn2(q, t)
    assert (i0[0], i1[0]) == (1, 2) and d0[0] == d1[0] == 1.0


def test_knn2_fewer_than_two_train_rows(oracle):
    q = np.zeros((2, 128), F32)
    i0, d0, i1, d1 = oracle.knn2(q, unit(0, 2)[None])
    assert list(i0) == [0, 0] and list(i1) == [-1, -1] and d1[0] == FLT_MAX
    i0, d0, i1, d1 = oracle.knn2(q, np.zeros((0, 128), F32))
    assert list(i0) == [-1, -1] and d0[0] == FLT_MAX


# ---- ratio / cross-check / distance filter ----------------------------------------------------

def test_ratio_is_strict_and_fp32(oracle):
    q = np.zeros((1, 128), F32)
    # d0 = 4, d1 = 5: 4 < 0.8f*5 ?  0.8f*5 = 4.0000000596.. -> rounds to 4.0 in fp32 -> not strictly less
    t = np.stack([unit(0, 4), unit(1, 5)])
    assert F32(0.8) * F32(5) == F32(4.0)
    assert len(oracle.compute_matches(q, t, 0.8)[0]) == 0
    t = np.stack([unit(0, 3.99), unit(1, 5)])
    mq, mt, md = oracle.compute_matches(q, t, 0.8)
    assert list(mq) == [0] and list(mt) == [0]


def test_cross_check_operator_bracket_quirk(oracle):
    # forward (0 -> 2) survives although train row 2 has NO reverse match: vis[2] default-inserts 0 == queryIdx 0
    q12, t12, d12 = np.array([0, 1, 3], np.int32), np.array([2, 5, 7], np.int32), np.array([.1, .2, .3], F32)
    q21, t21 = np.array([5, 7], np.int32), np.array([1, 9], np.int32)
    kq, kt, kd = oracle.cross_check((q12, t12, d12), (q21, t21))
    assert list(kq) == [0, 1] and list(kt) == [2, 5]
    # NumPy restatement agrees
    nq, nt, nd = no.cross_check((q12, t12, d12), (q21, t21))
    assert list(nq) == [0, 1] and list(nt) == [2, 5]


def test_distance_filter_boundary(oracle):
    d = np.array([0.7, np.nextafter(F32(0.7), F32(1)), 0.69999], F32)
    q = np.arange(3, dtype=np.int32)
    kq, _, _ = oracle.filter_by_distance((q, q, d), 0.7)
    assert list(kq) == [0, 2]  # 0.7f < 0.7 (double) is kept, the next float up is dropped


def test_match_pair_degenerate_sizes(oracle):
    A = np.random.default_rng(0).random((5, 128), dtype=F32)
    for n1, n2 in [(0, 5), (5, 0), (1, 5), (5, 1), (0, 0)]:
        assert len(oracle.match_pair(A[:n1], A[:n2])[0]) == 0
        assert len(no.match_pair(A[:n1], A[:n2])[0]) == 0


def test_self_match_is_identity(oracle):
    A = np.random.default_rng(1).random((64, 128), dtype=F32)
    q, t, d = oracle.match_pair(A, A, max_distance=10.0)
    assert np.array_equal(q, np.arange(64)) and np.array_equal(t, np.arange(64)) and (d == 0).all()


@pytest.mark.parametrize("order", [0, 1])
def test_match_pair_c_vs_numpy_on_clustered_data(oracle, order):
    from monocularsfm_amd import synth
    imgs = synth.rootsift_images(2, [420, 390], seed=3, n_proto=900)
    a = oracle.match_pair(imgs[0], imgs[1], order=order, nthreads=2)
    b = no.match_pair(imgs[0], imgs[1], order=order)
    assert len(a[0]) > 20
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(bits(a[2]), bits(b[2]))
    a = oracle.match_pair(imgs[0], imgs[1], cross_check=False, order=order)
    b = no.match_pair(imgs[0], imgs[1], do_cross_check=False, order=order)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- host logic ------------------------------------------------------------------------------

def test_pair_id_codec(oracle):
    assert oracle.pair_id(3, 7) == 3 * 10000 + 7 == oracle.pair_id(7, 3)
    assert oracle.pair_from_id(30007) == (3, 7)
    assert oracle.pair_id(0, 9999) == 9999 and oracle.pair_from_id(9999) == (0, 9999)


def test_enumerate_brute_order_and_batches(oracle):
    pairs, bend = oracle.enumerate_brute(5, 100)
    assert pairs.tolist() == [[1, 0], [2, 0], [2, 1], [3, 0], [3, 1], [3, 2], [4, 0], [4, 1], [4, 2], [4, 3]]
    assert bend.tolist() == [1, 3, 6, 10]  # flushed at the end of every row i
    pairs, bend = oracle.enumerate_brute(128, 100)
    assert len(pairs) == 8128 and (pairs[:, 0] > pairs[:, 1]).all()
    # row 127 has 127 pairs: one flush after 100, one at the end of the row
    assert bend[-2] == 8128 - 27 and bend[-1] == 8128
    pairs, bend = oracle.enumerate_brute(201, 100)
    assert bend[-1] == 201 * 200 // 2 and (np.diff(np.concatenate([[0], bend])) <= 100).all()


def test_enumerate_sequential(oracle):
    pairs, bend = oracle.enumerate_sequential(5, 3)
    assert pairs.tolist() == [[1, 0], [2, 1], [2, 0], [3, 2], [3, 1], [3, 0], [4, 3], [4, 2], [4, 1]]
    assert bend.tolist() == [1, 3, 6, 9]
    pairs, _ = oracle.enumerate_sequential(128, 3)
    assert len(pairs) == 378


def test_topscale_select(oracle):
    k = np.zeros((6, 4), F32)
    k[:, 2] = [2, 9, 9, 1, 7, 9]
    assert oracle.topscale_select(k, 4).tolist() == [1, 2, 5, 4]      # size desc, index asc on ties
    assert oracle.topscale_select(k, 6).tolist() == [1, 2, 5, 4, 0, 3]
    assert oracle.topscale_select(k, 7).tolist() == [0, 1, 2, 3, 4, 5]  # k > n: whole matrix, original order


def test_cache_blocked_loop_order_gives_identical_results(oracle):
    """orc_knn2_blocked (the pair-parallel CPU baseline's inner loop: 32 query rows x 128 train rows at a time) is a
    re-tiling of batchDistance's row-at-a-time loop, not a different computation: identical bits, ties included."""
    from monocularsfm_amd import synth
    imgs = synth.rootsift_images(2, [333, 415], seed=31, n_proto=500)
    u = synth.u8_images(2, [200, 257], seed=32, dup_frac=0.3)
    u[1][7] = u[1][3]
    u[1][140] = u[1][3]
    u[0][5] = u[1][3]
    for A, B in ((imgs[0], imgs[1]), (u[0], u[1]), (imgs[0][:1], imgs[1][:1]), (imgs[0][:40], imgs[1][:129]), (imgs[0][:3], imgs[1][:0])):
        for order in (0, 1, 2):
            for x, y in zip(oracle.knn2(A, B, order), oracle.knn2_blocked(A, B, order)):
                assert np.array_equal(x.view(np.int32), y.view(np.int32))
    # and the pair-parallel entry point (which uses it) equals the per-pair calls
    offs, q, t, d = oracle.match_pairs([imgs[0], imgs[1], u[0], u[1]], [(0, 1), (1, 0), (2, 3)], max_distance=1e9, nthreads=3)
    for p, (i, j) in enumerate([(0, 1), (1, 0), (2, 3)]):
        oq, ot, od = oracle.match_pair([imgs[0], imgs[1], u[0], u[1]][i], [imgs[0], imgs[1], u[0], u[1]][j], max_distance=1e9)
        assert np.array_equal(q[offs[p]:offs[p + 1]], oq) and np.array_equal(t[offs[p]:offs[p + 1]], ot)
        assert np.array_equal(d[offs[p]:offs[p + 1]].view(np.int32), od.view(np.int32))
