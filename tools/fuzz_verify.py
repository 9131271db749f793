"""Seeded fuzz of the batched device RANSAC against its host twin on random / degenerate correspondences
(identical masks required).  Usage: python tools/fuzz_verify.py [seed] [cases]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth
F32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
host = C.CDLL(os.path.join(ROOT, "monocularsfm_amd", "host", "libmsfm_host.so"))
fp = C.POINTER(C.c_float)

def host_mask(p1, p2):
    p1 = np.ascontiguousarray(p1, F32); p2 = np.ascontiguousarray(p2, F32)
    mask = np.zeros(max(len(p1), 1), np.uint8)
    n = host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), len(p1), mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return mask[:n].astype(bool) if n else np.zeros(len(p1), bool)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
ctx = _lib.Context(0)
bad = 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for case in range(cases):
    n_pairs = int(rng.integers(1, 5))
    descs, kps = [], []
    for p in range(n_pairs):
        n = int(rng.choice([0, 5, 7, 8, 9, 15, 40, 120, 400, 900]))
        extra = int(rng.integers(5, 60))
        base = synth.rootsift_images(1, [n + 2 * extra], seed=100 * case + p, n_proto=4 * (n + 2 * extra) + 64)[0]
        dA = np.r_[base[:n], base[n:n + extra]]
        nb = np.abs(base[:n] + rng.normal(0, 0.003, (n, 128)).astype(F32)); nb /= np.maximum(np.linalg.norm(nb, axis=1, keepdims=True), 1e-9)
        dB = np.r_[nb.astype(F32), base[n + extra:]]
        mode = rng.choice(["geometry", "random", "collinear", "samepoint", "duplicates"])
        m = len(dA)
        if mode == "geometry":
            n_in = int(n * rng.uniform(0.3, 1.0))
            X = np.c_[rng.uniform(-2, 2, n_in), rng.uniform(-1.5, 1.5, n_in), rng.uniform(4, 9, n_in)]
            K = np.array([[2559.68, 0, 1536], [0, 2559.68, 1152], [0, 0, 1]]); a = 0.15
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]); t = np.array([0.8, 0.05, 0.1])
            x1 = (K @ X.T).T; x1 = x1[:, :2] / x1[:, 2:]
            x2 = (K @ (R @ X.T + t[:, None])).T; x2 = x2[:, :2] / x2[:, 2:]
            kA = np.r_[x1 + rng.normal(0, 0.5, x1.shape), np.c_[rng.uniform(0, 3072, m - n_in), rng.uniform(0, 2304, m - n_in)]]
            kB = np.r_[x2 + rng.normal(0, 0.5, x2.shape), np.c_[rng.uniform(0, 3072, m - n_in), rng.uniform(0, 2304, m - n_in)]]
        elif mode == "random":
            kA = np.c_[rng.uniform(0, 3072, m), rng.uniform(0, 2304, m)]; kB = np.c_[rng.uniform(0, 3072, m), rng.uniform(0, 2304, m)]
        elif mode == "collinear":
            s = rng.uniform(0, 3000, m); kA = np.c_[s, 0.5 * s + 10]; kB = np.c_[s + 5, 0.5 * s + 30]
        elif mode == "samepoint":
            kA = np.tile([[100.0, 200.0]], (m, 1)); kB = np.tile([[300.0, 50.0]], (m, 1))
        else:
            kA = np.c_[rng.integers(0, 20, m) * 100.0, rng.integers(0, 20, m) * 100.0]; kB = kA + 3.0
        descs += [dA.astype(F32), dB.astype(F32)]
        kps += [np.c_[kA, np.ones(m), np.zeros(m)].astype(F32), np.c_[kB, np.ones(m), np.zeros(m)].astype(F32)]
    for i, (d, k) in enumerate(zip(descs, kps)):
        ctx.upload_image(i, d); ctx.upload_keypoints(i, k)
    pairs = np.array([(2 * p, 2 * p + 1) for p in range(n_pairs)] + [(1, 0)], np.int32)
    thr = float(rng.choice([0.5, 3.0, 10.0])); iters = int(rng.choice([50, 1000]))
    offs, qt, d = ctx.match_pairs(pairs)
    voffs, vqt, vd = ctx.match_pairs_verified(pairs, threshold=thr, max_iters=iters)
    ok = True
    for p, (i, j) in enumerate(pairs):
        s, e = offs[p], offs[p + 1]
        q, t = qt[s:e, 0], qt[s:e, 1]
        if thr == 3.0 and iters == 1000:
            keep = host_mask(kps[i][q, :2], kps[j][t, :2]) if e > s else np.zeros(0, bool)
            exp = qt[s:e][keep]
            got = vqt[voffs[p]:voffs[p + 1]]
            ok &= np.array_equal(exp, got)
        else:
            got = vqt[voffs[p]:voffs[p + 1]]
            ok &= set(map(tuple, got.tolist())) <= set(map(tuple, qt[s:e].tolist()))
    if not ok:
        bad += 1
        print("MISMATCH case", case, flush=True)
print("cases done, mismatches:", bad)
