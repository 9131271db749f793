"""Independent NumPy float32 restatement of the matcher, used to cross-check the C oracle.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/__init__.py).  Written from the
behavioural spec, not from msfm_oracle.c: a bug has to be made twice,
independently, to go unnoticed.  NumPy float32 elementwise ops are single-rounded IEEE ops
(no FMA), which is what the SSE4X4 and SCALAR orders need; the AVX2_FMA order is emulated
through float64 (the product of two float32 is exact in float64; double rounding of the
fused sum is possible in principle but needs a 29-bit coincidence -- the C oracle uses fmaf).
"""
import numpy as np

ORDER_SSE4X4 = 0
ORDER_AVX2_FMA = 1
ORDER_SCALAR = 2
ORDER_AVX512_FMA = 3
F32 = np.float32


def l2sqr_matrix(A, B, order=ORDER_SSE4X4):
    """S[q][t] in float32 with the named accumulation order.  A: n1x128, B: n2x128."""
    A = np.asarray(A, dtype=F32)
    B = np.asarray(B, dtype=F32)
    n1, n2 = A.shape[0], B.shape[0]
    T = A[:, None, :] - B[None, :, :]  # float32 subtract
    if order == ORDER_SSE4X4:
        SQ = T * T  # float32 multiply, rounded
        SQ = SQ.reshape(n1, n2, 8, 16)  # [it][L], k = 16*it + L
        p = SQ[:, :, 0, :].copy()
        for it in range(1, 8):
            p = p + SQ[:, :, it, :]
        s = ((p[..., 0:4] + p[..., 4:8]) + p[..., 8:12]) + p[..., 12:16]
        return (s[..., 0] + s[..., 2]) + (s[..., 1] + s[..., 3])
    if order == ORDER_SCALAR:
        SQ = T * T
        d = np.zeros((n1, n2), F32)
        for c in range(128):
            d = d + SQ[:, :, c]
        return d
    if order == ORDER_AVX2_FMA:
        T64 = T.astype(np.float64).reshape(n1, n2, 4, 32)
        p = np.zeros((n1, n2, 32), F32)
        for it in range(4):
            p = (T64[:, :, it, :] * T64[:, :, it, :] + p.astype(np.float64)).astype(F32)
        s = ((p[..., 0:8] + p[..., 8:16]) + p[..., 16:24]) + p[..., 24:32]
        return ((s[..., 0] + s[..., 1]) + (s[..., 2] + s[..., 3])) + ((s[..., 4] + s[..., 5]) + (s[..., 6] + s[..., 7]))
    if order == ORDER_AVX512_FMA:
        T64 = T.astype(np.float64).reshape(n1, n2, 2, 64)
        p = np.zeros((n1, n2, 64), F32)
        for it in range(2):   # (a float32 product of float32 values is exact in float64: one rounding, like fmaf)
            p = (T64[:, :, it, :] * T64[:, :, it, :] + p.astype(np.float64)).astype(F32)
        s = ((p[..., 0:16] + p[..., 16:32]) + p[..., 32:48]) + p[..., 48:64]
        y = (s[..., 0:4] + s[..., 8:12]) + (s[..., 4:8] + s[..., 12:16])
        return (y[..., 0] + y[..., 2]) + (y[..., 1] + y[..., 3])
    raise ValueError(order)


def knn2(A, B, order=ORDER_SSE4X4, block=64):
    """Two nearest train rows per query row under (sqrt distance asc, index asc)."""
    A = np.asarray(A, dtype=F32)
    B = np.asarray(B, dtype=F32)
    n1, n2 = A.shape[0], B.shape[0]
    idx0 = np.full(n1, -1, np.int32)
    idx1 = np.full(n1, -1, np.int32)
    d0 = np.full(n1, np.finfo(F32).max, F32)
    d1 = np.full(n1, np.finfo(F32).max, F32)
    if n2 == 0:
        return idx0, d0, idx1, d1
    for s in range(0, n1, block):
        D = np.sqrt(l2sqr_matrix(A[s:s + block], B, order)).astype(F32)
        # stable argsort == ties broken by lowest train index
        o = np.argsort(D, axis=1, kind="stable")
        r = np.arange(D.shape[0])
        idx0[s:s + block] = o[:, 0]
        d0[s:s + block] = D[r, o[:, 0]]
        if n2 >= 2:
            idx1[s:s + block] = o[:, 1]
            d1[s:s + block] = D[r, o[:, 1]]
    return idx0, d0, idx1, d1


def compute_matches(A, B, ratio=0.8, order=ORDER_SSE4X4):
    A = np.asarray(A, dtype=F32)
    B = np.asarray(B, dtype=F32)
    if A.shape[0] == 0 or B.shape[0] < 2:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, F32)
    i0, d0, _, d1 = knn2(A, B, order)
    keep = d0 < (F32(ratio) * d1)
    q = np.nonzero(keep)[0].astype(np.int32)
    return q, i0[q], d0[q]


def cross_check(m12, m21):
    q12, t12, d12 = m12
    vis = {}
    for q, t in zip(m21[0].tolist(), m21[1].tolist()):
        vis[q] = t
    keep = [i for i, (q, t) in enumerate(zip(q12.tolist(), t12.tolist())) if vis.get(t, 0) == q]
    keep = np.asarray(keep, dtype=np.int64)
    return q12[keep], t12[keep], d12[keep]


def match_pair(A, B, ratio=0.8, do_cross_check=True, max_distance=0.7, order=ORDER_SSE4X4):
    A = np.asarray(A, dtype=F32)
    B = np.asarray(B, dtype=F32)
    m = compute_matches(A, B, ratio, order)
    if do_cross_check:
        if A.shape[0] < 2 or B.shape[0] < 2:
            return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, F32)
        m = cross_check(m, compute_matches(B, A, ratio, order))
    q, t, d = m
    keep = ~(d.astype(np.float64) > float(max_distance))
    return q[keep], t[keep], d[keep]
