"""ctypes binding of oracle/libmsfm_oracle.so (the C restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmsfm_oracle.so")

ORDER_SSE4X4 = 0
ORDER_AVX2_FMA = 1
ORDER_SCALAR = 2
ORDER_AVX512_FMA = 3
ORDER_SSE4X4_PLAINC = 100  # test hook: plain-C statement of the SSE order


def build(force=False):
    src = os.path.join(_HERE, "msfm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        lp = C.POINTER(C.c_int64)
        L.orc_l2sqr.restype = C.c_float
        L.orc_l2sqr.argtypes = [fp, fp, C.c_int]
        L.orc_knn2_mt.restype = None
        L.orc_knn2_mt.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, ip, fp, ip, fp]
        L.orc_knn2_blocked.restype = None
        L.orc_knn2_blocked.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, ip, fp, ip, fp]
        L.orc_compute_matches.restype = C.c_int
        L.orc_compute_matches.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float, C.c_int, C.c_int, ip, ip, fp]
        L.orc_cross_check.restype = C.c_int
        L.orc_cross_check.argtypes = [ip, ip, fp, C.c_int, ip, ip, C.c_int, ip, ip, fp]
        L.orc_filter_by_distance.restype = C.c_int
        L.orc_filter_by_distance.argtypes = [ip, ip, fp, C.c_int, C.c_double, ip, ip, fp]
        L.orc_match_pair.restype = C.c_int
        L.orc_match_pair.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float, C.c_int, C.c_double,
                                     C.c_int, C.c_int, ip, ip, fp]
        L.orc_match_pairs_mt.restype = C.c_int64
        L.orc_match_pairs_mt.argtypes = [C.POINTER(fp), ip, ip, C.c_int, C.c_float, C.c_int, C.c_double, C.c_int, C.c_int,
                                         C.c_double, C.POINTER(C.c_int), lp, ip, ip, fp]
        L.orc_topscale_select.restype = C.c_int
        L.orc_topscale_select.argtypes = [fp, C.c_int, C.c_int, ip]
        L.orc_pair_id.restype = C.c_int32
        L.orc_pair_id.argtypes = [C.c_int32, C.c_int32]
        L.orc_pair_from_id.restype = None
        L.orc_pair_from_id.argtypes = [C.c_int32, ip, ip]
        L.orc_swap_image_pair.restype = C.c_int
        L.orc_swap_image_pair.argtypes = [C.c_int32, C.c_int32]
        L.orc_enumerate_brute.restype = C.c_int64
        L.orc_enumerate_brute.argtypes = [C.c_int, C.c_int, ip, lp, lp]
        L.orc_enumerate_sequential.restype = C.c_int64
        L.orc_enumerate_sequential.argtypes = [C.c_int, C.c_int, ip, lp, lp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _desc(d):
    d = np.ascontiguousarray(d, dtype=np.float32)
    if d.ndim != 2 or (d.shape[0] and d.shape[1] != 128):
        raise ValueError("descriptors must be n x 128")
    return d


def l2sqr(a, b, order=ORDER_SSE4X4):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_l2sqr(_f(a), _f(b), order))


def knn2(q, t, order=ORDER_SSE4X4, nthreads=1):
    """-> idx0, d0, idx1, d1 (int32, float32) like knnMatch(q, t, k=2)."""
    q, t = _desc(q), _desc(t)
    nq, nt = q.shape[0], t.shape[0]
    idx0 = np.empty(nq, np.int32)
    idx1 = np.empty(nq, np.int32)
    d0 = np.empty(nq, np.float32)
    d1 = np.empty(nq, np.float32)
    lib().orc_knn2_mt(_f(q), nq, _f(t), nt, order, nthreads, _i(idx0), _f(d0), _i(idx1), _f(d1))
    return idx0, d0, idx1, d1


def knn2_blocked(q, t, order=ORDER_SSE4X4):
    """knn2 from the cache-blocked loop order (the pair-parallel baseline's inner loop): identical results."""
    q, t = _desc(q), _desc(t)
    nq, nt = q.shape[0], t.shape[0]
    idx0 = np.empty(nq, np.int32)
    idx1 = np.empty(nq, np.int32)
    d0 = np.empty(nq, np.float32)
    d1 = np.empty(nq, np.float32)
    lib().orc_knn2_blocked(_f(q), nq, _f(t), nt, order, _i(idx0), _f(d0), _i(idx1), _f(d1))
    return idx0, d0, idx1, d1


def compute_matches(d1, d2, ratio=0.8, order=ORDER_SSE4X4, nthreads=1):
    d1, d2 = _desc(d1), _desc(d2)
    n1 = d1.shape[0]
    q = np.empty(max(n1, 1), np.int32)
    t = np.empty(max(n1, 1), np.int32)
    d = np.empty(max(n1, 1), np.float32)
    m = lib().orc_compute_matches(_f(d1), n1, _f(d2), d2.shape[0], np.float32(ratio), order,
                                  nthreads, _i(q), _i(t), _f(d))
    return q[:m].copy(), t[:m].copy(), d[:m].copy()


def cross_check(m12, m21):
    q12, t12, d12 = [np.ascontiguousarray(x) for x in m12]
    q21, t21 = [np.ascontiguousarray(x, dtype=np.int32) for x in m21[:2]]
    q12 = q12.astype(np.int32)
    t12 = t12.astype(np.int32)
    d12 = d12.astype(np.float32)
    n = max(len(q12), 1)
    q = np.empty(n, np.int32)
    t = np.empty(n, np.int32)
    d = np.empty(n, np.float32)
    m = lib().orc_cross_check(_i(q12), _i(t12), _f(d12), len(q12), _i(q21), _i(t21), len(q21),
                              _i(q), _i(t), _f(d))
    return q[:m].copy(), t[:m].copy(), d[:m].copy()


def filter_by_distance(m, max_distance=0.7):
    q0, t0, d0 = m
    q0 = np.ascontiguousarray(q0, dtype=np.int32)
    t0 = np.ascontiguousarray(t0, dtype=np.int32)
    d0 = np.ascontiguousarray(d0, dtype=np.float32)
    n = max(len(q0), 1)
    q = np.empty(n, np.int32)
    t = np.empty(n, np.int32)
    d = np.empty(n, np.float32)
    k = lib().orc_filter_by_distance(_i(q0), _i(t0), _f(d0), len(q0), float(max_distance),
                                     _i(q), _i(t), _f(d))
    return q[:k].copy(), t[:k].copy(), d[:k].copy()


def match_pair(d1, d2, ratio=0.8, cross_check=True, max_distance=0.7, order=ORDER_SSE4X4, nthreads=1):
    """ComputeCrossMatches/ComputeMatches + FilterMatchesByDistance -> (q, t, dist)."""
    d1, d2 = _desc(d1), _desc(d2)
    n1 = d1.shape[0]
    q = np.empty(max(n1, 1), np.int32)
    t = np.empty(max(n1, 1), np.int32)
    d = np.empty(max(n1, 1), np.float32)
    m = lib().orc_match_pair(_f(d1), n1, _f(d2), d2.shape[0], np.float32(ratio), int(bool(cross_check)),
                             float(max_distance), order, nthreads, _i(q), _i(t), _f(d))
    return q[:m].copy(), t[:m].copy(), d[:m].copy()


def match_pairs(images, pairs, ratio=0.8, cross_check=True, max_distance=0.7, order=ORDER_SSE4X4, nthreads=None,
                budget_s=0.0, return_done=False):
    """MatchImagePairs' loop over independent pairs, parallel over PAIRS (persistent pool, one pair per worker at a
    time).  images: list / dict id -> n x 128 float32; pairs: P x 2 (query id, train id).
    -> offsets int64[P+1], q int32[M], t int32[M], dist float32[M].  budget_s > 0 stops starting new pairs after that
    many seconds; with return_done the number of pairs computed (a prefix of the list) is returned as a fifth value."""
    pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
    P = pairs.shape[0]
    ids = sorted(set(pairs.ravel().tolist()))
    n_slots = (max(ids) + 1) if ids else 1
    keep = {i: _desc(images[i]) for i in ids}
    ptrs = (C.POINTER(C.c_float) * n_slots)()
    rows = np.zeros(n_slots, np.int32)
    for i, d in keep.items():
        ptrs[i] = _f(d)
        rows[i] = d.shape[0]
    cap = int(rows[pairs[:, 0]].sum()) if P else 0
    q = np.empty(max(cap, 1), np.int32)
    t = np.empty(max(cap, 1), np.int32)
    d = np.empty(max(cap, 1), np.float32)
    offs = np.zeros(P + 1, np.int64)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    done = C.c_int()
    m = lib().orc_match_pairs_mt(ptrs, _i(rows), _i(pairs), P, np.float32(ratio), int(bool(cross_check)), float(max_distance),
                                 order, int(nthreads), float(budget_s), C.byref(done),
                                 offs.ctypes.data_as(C.POINTER(C.c_int64)), _i(q), _i(t), _f(d))
    if return_done:
        return offs, q[:m].copy(), t[:m].copy(), d[:m].copy(), done.value
    return offs, q[:m].copy(), t[:m].copy(), d[:m].copy()


def topscale_select(kpts, k):
    kpts = np.ascontiguousarray(kpts, dtype=np.float32).reshape(-1, 4)
    n = kpts.shape[0]
    out = np.empty(max(n, k, 1), np.int32)
    m = lib().orc_topscale_select(_f(kpts), n, k, _i(out))
    return out[:m].copy()


def pair_id(id1, id2):
    return int(lib().orc_pair_id(id1, id2))


def pair_from_id(pid):
    a = C.c_int32()
    b = C.c_int32()
    lib().orc_pair_from_id(pid, C.byref(a), C.byref(b))
    return a.value, b.value


def enumerate_brute(n_images, max_pairs=100):
    npairs = n_images * (n_images - 1) // 2
    pairs = np.empty((max(npairs, 1), 2), np.int32)
    bend = np.empty(max(npairs, 1) + n_images + 1, np.int64)
    nb = C.c_int64()
    n = lib().orc_enumerate_brute(n_images, max_pairs, _i(pairs),
                                  bend.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nb))
    return pairs[:n].copy(), bend[:nb.value].copy()


def enumerate_sequential(n_images, overlap=3):
    cap = max(n_images * overlap, 1)
    pairs = np.empty((cap, 2), np.int32)
    bend = np.empty(max(n_images, 1), np.int64)
    nb = C.c_int64()
    n = lib().orc_enumerate_sequential(n_images, overlap, _i(pairs),
                                       bend.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nb))
    return pairs[:n].copy(), bend[:nb.value].copy()
