"""Seeded fuzz of the device results against the CPU oracle (test infrastructure): random sizes, value types,
duplicates, parameters and both accumulation orders.  Usage: python tools/fuzz_oracle.py [seed] [cases]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth
from oracle import c_oracle as oracle
F32 = np.float32
b = lambda a: np.asarray(a).view(np.int32) if np.asarray(a).dtype == np.float32 else np.asarray(a)
ctx = _lib.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    n1, n2 = (int(rng.choice([1, 2, 3, 17, 64, 129, 300, 700, 1500])) for _ in range(2))
    kind = rng.choice(["rootsift", "u8", "gauss"])
    if kind == "rootsift":
        A, B = synth.rootsift_images(2, [n1, n2], seed=9000 + case, n_proto=max(n1, n2) + 40)
    elif kind == "u8":
        A, B = (x.astype(F32) for x in synth.u8_images(2, [n1, n2], seed=9500 + case, as_float=True))
    else:
        A, B = rng.normal(size=(n1, 128)).astype(F32), rng.normal(size=(n2, 128)).astype(F32)
    if rng.random() < 0.4 and n1 >= 2 and n2 >= 2:
        B[n2 // 2] = A[0]; A[-1] = A[0]
    order = int(rng.integers(0, 2)); ratio = float(rng.choice([0.5, 0.8, 1.0])); cc = bool(rng.integers(0, 2))
    md = float(rng.choice([0.3, 0.7, 1e9]))
    pf = bool(rng.integers(0, 2))
    ctx.set_accum_order(order); ctx.set_prefilter(pf)
    ctx.upload_image(0, A); ctx.upload_image(1, B)
    q, t, d = ctx.match_pair(0, 1, ratio, cc, md)
    fwd, rev = ctx.knn2_pair(0, 1)
    oq, ot, od = oracle.match_pair(A, B, ratio, cc, md, order, 4)
    oi0, od0, _, od1 = oracle.knn2(A, B, order, 4)
    ok = np.array_equal(q, oq) and np.array_equal(t, ot) and np.array_equal(b(d), b(od))
    ok &= np.array_equal(fwd[0], oi0) and np.array_equal(b(fwd[1]), b(od0)) and np.array_equal(b(fwd[2]), b(od1))
    if not ok:
        bad += 1
        print("MISMATCH", case, kind, n1, n2, order, ratio, cc, md, pf, flush=True)
ctx.set_prefilter(True); ctx.set_accum_order(0)
print("cases done, mismatches:", bad)
