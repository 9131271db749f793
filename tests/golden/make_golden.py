"""Generates tests/golden/*.npz -- inputs and expected outputs for the ComputeMatches hot path.

PROVENANCE: the reference holds no golden vectors for this path and its arithmetic (OpenCV
BFMatcher) cannot be run here, so the expected outputs below are produced by THIS build's CPU
oracle (oracle/msfm_oracle.c), cross-checked against the independent NumPy restatement before
being written.  They are regression anchors for oracle and HIP path alike, not OpenCV outputs
("parity unpinned").  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from monocularsfm_amd import synth  # noqa: E402
from oracle import c_oracle as co, np_oracle as no  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


def case(name, A, B, orders=(0, 1)):
    out = {"desc1": A, "desc2": B}
    for o in orders:
        f = co.knn2(A, B, o)
        r = co.knn2(B, A, o)
        nf, nr = no.knn2(A, B, o), no.knn2(B, A, o)
        for x, y in zip(f + r, nf + nr):
            assert np.array_equal(x.view(np.int32), y.view(np.int32)), "C and NumPy oracle disagree"
        out["o%d_fwd_idx0" % o], out["o%d_fwd_d0" % o], _, out["o%d_fwd_d1" % o] = f
        out["o%d_rev_idx0" % o], out["o%d_rev_d0" % o], _, out["o%d_rev_d1" % o] = r
        for cc in (1, 0):
            q, t, d = co.match_pair(A, B, 0.8, bool(cc), 0.7 if A.max() <= 1.5 else 1e9, o)
            nq, nt, nd = no.match_pair(A, B, 0.8, bool(cc), 0.7 if A.max() <= 1.5 else 1e9, o)
            assert np.array_equal(q, nq) and np.array_equal(t, nt)
            out["o%d_cc%d_q" % (o, cc)], out["o%d_cc%d_t" % (o, cc)], out["o%d_cc%d_d" % (o, cc)] = q, t, d
    out["max_distance"] = np.float64(0.7 if A.max() <= 1.5 else 1e9)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, A.shape, B.shape, {k: len(v) for k, v in out.items() if k.endswith("_q")})


def main():
    imgs = synth.rootsift_images(2, [200, 180], seed=101, n_proto=380)
    case("rootsift_200x180", imgs[0], imgs[1])

    u = synth.u8_images(2, [150, 161], seed=102, dup_frac=0.2)
    u[1][7] = u[1][3]       # exact duplicate train rows -> ties
    u[1][90] = u[1][3]
    u[0][10] = u[1][3]      # and a query identical to them (distance 0 three times)
    u[0][0] = u[1][40]      # query 0 has an exact twin: exercises the queryIdx==0 cross-check quirk path
    case("u8_ties_150x161", u[0], u[1])

    # ragged: non multiples of the 128-row tile, tiny second image
    imgs = synth.rootsift_images(2, [131, 5], seed=103, n_proto=140)
    case("ragged_131x5", imgs[0], imgs[1])

    # sqrt-space tie: distinct S, equal sqrtf(S); lower index has the larger S
    A = np.zeros((3, 128), F32)
    B = np.zeros((4, 128), F32)
    B[0, 0] = 9.0
    B[1, 0], B[1, 1] = 1.0, 2.0 ** -11.5
    B[2, 0] = 1.0
    B[3, 5] = 3.0
    A[1, 0] = 0.5
    A[2, 5] = 2.0
    case("sqrt_tie_3x4", A, B)

    # CrossCheck operator[] quirk (FeatureUtils.cpp:302): forward (0 -> 5) survives although train
    # row 5 has no reverse match (its two nearest queries are near-twins, so the reverse ratio test
    # fails); the same situation for query 3 -> train 6 is dropped.
    rng = np.random.default_rng(104)
    imgs = synth.rootsift_images(2, [40, 30], seed=104, n_proto=4000, overlap=0.0)
    A, B = imgs[0].copy(), imgs[1].copy()

    def jitter(v, s):
        w = np.abs(v + s * rng.standard_normal(128).astype(F32) * v)
        return (w / np.linalg.norm(w)).astype(F32)
    A[7] = jitter(A[0], 0.002)
    A[9] = jitter(A[3], 0.002)
    B[5] = jitter(A[0], 0.05)
    B[6] = jitter(A[3], 0.05)
    case("quirk_40x30", A, B)
    g = np.load(os.path.join(HERE, "quirk_40x30.npz"))
    cc1 = set(zip(g["o0_cc1_q"].tolist(), g["o0_cc1_t"].tolist()))
    cc0 = set(zip(g["o0_cc0_q"].tolist(), g["o0_cc0_t"].tolist()))
    assert {(0, 5), (7, 5), (3, 6), (9, 6)} <= cc0, cc0
    assert (0, 5) in cc1 and (7, 5) not in cc1 and (3, 6) not in cc1 and (9, 6) not in cc1, cc1


def int_case():
    """u8_int64_400x380.npz: expected outputs from the EXACT-INTEGER reference (oracle/int_oracle.py), i.e. what every
    conforming OpenCV build returns for integer descriptors -- not from the restated SIMD order.  The C oracle must
    agree under every order before the file is written; int_fwd_s0 / s1 are the integer S of both neighbours."""
    from oracle import int_oracle as io
    u = synth.u8_images(2, [400, 380], seed=105, dup_frac=0.15)
    A, B = u[0], u[1]
    B[7] = B[3]
    B[90] = B[3]
    A[10] = B[3]
    A[0] = B[40]
    out = {"desc1": A.astype(np.uint8), "desc2": B.astype(np.uint8), "max_distance": np.float64(1e9)}
    f, r = io.knn2(A, B), io.knn2(B, A)
    for o in (0, 1, 2):
        for x, y in zip(f + r, co.knn2(A, B, o) + co.knn2(B, A, o)):
            assert np.array_equal(x.view(np.int32), y.view(np.int32)), "C oracle != integer reference"
    S = np.sort(io.s_matrix(A, B), axis=1)
    out["int_fwd_s0"], out["int_fwd_s1"] = S[:, 0], S[:, 1]
    for o in (0, 1):
        out["o%d_fwd_idx0" % o], out["o%d_fwd_d0" % o], _, out["o%d_fwd_d1" % o] = f
        out["o%d_rev_idx0" % o], out["o%d_rev_d0" % o], _, out["o%d_rev_d1" % o] = r
        for cc in (1, 0):
            out["o%d_cc%d_q" % (o, cc)], out["o%d_cc%d_t" % (o, cc)], out["o%d_cc%d_d" % (o, cc)] = io.match_pair(A, B, 0.8, bool(cc), 1e9)
    np.savez_compressed(os.path.join(HERE, "u8_int64_400x380.npz"), **out)
    print("u8_int64_400x380", {k: len(v) for k, v in out.items() if k.endswith("_q")})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "int":
        int_case()
    else:
        main()
        int_case()
