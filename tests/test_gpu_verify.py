"""Batched device RANSAC (msfm_match_pairs_verified = matching + FeatureUtils::FilterMatches) against its host
twin (host/GeometricVerification.cpp through libmsfm_host.so): the two share the fp64 arithmetic of
csrc/msfm_fmat.h, the sampling stream and the adaptive stopping rule, so the verified lists must be IDENTICAL;
on data with a true epipolar geometry the inliers must be recovered.  (Not a parity claim against OpenCV's
findFundamentalMat -- SURVEY.md 8a-a13.)"""
import ctypes as C
import os

import numpy as np
import pytest

from monocularsfm_amd import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(built_lib):
    L = C.CDLL(os.path.join(ROOT, "monocularsfm_amd", "host", "libmsfm_host.so"))
    L.host_fundamental_ransac.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_ubyte)]
    return L


def host_mask(host, p1, p2):
    p1 = np.ascontiguousarray(p1, F32)
    p2 = np.ascontiguousarray(p2, F32)
    mask = np.zeros(max(len(p1), 1), np.uint8)
    fp = C.POINTER(C.c_float)
    n = host.host_fundamental_ransac(p1.ctypes.data_as(fp), p2.ctypes.data_as(fp), len(p1), mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return mask[:n].astype(bool) if n else np.zeros(len(p1), bool)


def two_view_scene(n_in, n_out, n_extra, seed, noise=0.4):
    """Two images whose descriptors match one-to-one on the first n_in + n_out rows (in shuffled order); the
    first n_in correspondences obey one epipolar geometry, the n_out others have random keypoints."""
    rng = np.random.default_rng(seed)
    n = n_in + n_out
    base = synth.rootsift_images(1, [n + 2 * n_extra], seed=seed, n_proto=4 * (n + 2 * n_extra) + 64)[0]
    dA = np.r_[base[:n], base[n:n + n_extra]]
    nb = np.abs(base[:n] + rng.normal(0, 0.004, (n, 128)).astype(F32))
    nb /= np.linalg.norm(nb, axis=1, keepdims=True)
    dB = np.r_[nb.astype(F32), base[n + n_extra:]]
    X = np.c_[rng.uniform(-2, 2, n_in), rng.uniform(-1.5, 1.5, n_in), rng.uniform(4, 9, n_in)]
    K = np.array([[2559.68, 0, 1536], [0, 2559.68, 1152], [0, 0, 1]])
    a = 0.1 + 0.1 * rng.random()
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.8, 0.05, 0.1]) * (0.5 + rng.random())
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:]
    x2 = (K @ (R @ X.T + t[:, None])).T
    x2 = x2[:, :2] / x2[:, 2:]
    x1 += rng.normal(0, noise, x1.shape)
    x2 += rng.normal(0, noise, x2.shape)
    rnd = lambda m: np.c_[rng.uniform(0, 3072, m), rng.uniform(0, 2304, m)]
    kA = np.r_[x1, rnd(n_out + n_extra)]
    kB = np.r_[x2, rnd(n_out + n_extra)]
    pa, pb = rng.permutation(len(dA)), rng.permutation(len(dB))
    kpA = np.c_[kA[pa], np.full(len(pa), 3.0), np.zeros(len(pa))].astype(F32)   # Database layout: x, y, size, angle
    kpB = np.c_[kB[pb], np.full(len(pb), 3.0), np.zeros(len(pb))].astype(F32)
    true_rows_A = np.zeros(len(dA), bool)
    true_rows_A[:n_in] = True
    return dA[pa].astype(F32), kpA, dB[pb].astype(F32), kpB, true_rows_A[pa]


def expected_lists(ctx, host, pairs, kps, **kw):
    offs, qt, d = ctx.match_pairs(pairs, **kw)
    out_q, out_d, out_off = [], [], [0]
    for p, (i, j) in enumerate(pairs):
        s, e = offs[p], offs[p + 1]
        q, t = qt[s:e, 0], qt[s:e, 1]
        keep = host_mask(host, kps[i][q, :2], kps[j][t, :2]) if e > s else np.zeros(0, bool)
        out_q.append(qt[s:e][keep])
        out_d.append(d[s:e][keep])
        out_off.append(out_off[-1] + int(keep.sum()))
    return np.asarray(out_off, np.int64), np.concatenate(out_q).reshape(-1, 2), np.concatenate(out_d), (offs, qt)


def test_verified_lists_equal_the_host_twin_and_recover_the_geometry(gpu_ctx, host):
    scenes = [two_view_scene(300, 120, 200, seed=3), two_view_scene(200, 120, 100, seed=4), two_view_scene(900, 50, 50, seed=5, noise=1.0)]
    descs, kps, truth = [], [], []
    for dA, kA, dB, kB, tr in scenes:
        descs += [dA, dB]
        kps += [kA, kB]
        truth.append(tr)
    for i, (dd, kk) in enumerate(zip(descs, kps)):
        gpu_ctx.upload_image(i, dd)
        gpu_ctx.upload_keypoints(i, kk)
    # the three true pairs (both orientations of the first), plus unrelated pairs (no consensus expected)
    pairs = np.array([(0, 1), (1, 0), (2, 3), (4, 5), (0, 3), (2, 5), (4, 1)], np.int32)
    exp_off, exp_qt, exp_d, (raw_off, raw_qt) = expected_lists(gpu_ctx, host, pairs, kps)
    offs, qt, d = gpu_ctx.match_pairs_verified(pairs)
    prof = gpu_ctx.profile()
    assert np.array_equal(offs, exp_off) and np.array_equal(qt, exp_qt) and np.array_equal(d.view(np.int32), exp_d.view(np.int32))
    assert prof["verify_ms"] > 0
    # geometry: the true correspondences survive, the planted false ones do not
    for p, sc in ((0, 0), (2, 1), (3, 2)):
        q = qt[offs[p]:offs[p + 1], 0]
        rq = raw_qt[raw_off[p]:raw_off[p + 1], 0]
        tr = truth[sc]
        assert tr[q].sum() >= 0.9 * tr[rq].sum() > 20
        assert (~tr[q]).sum() <= 0.15 * max(1, (~tr[rq]).sum())
    for p in (4, 5, 6):   # unrelated images: whatever matched by accident has no common geometry
        assert offs[p + 1] - offs[p] <= 12


@pytest.mark.parametrize("n_match", [0, 3, 6, 7, 8, 9, 20])
def test_small_match_counts_follow_findFundamentalMat_cases(gpu_ctx, host, n_match):
    """0 matches -> nothing; < 7 -> no model -> nothing; exactly 7 -> all kept; >= 8 -> RANSAC."""
    dA, kA, dB, kB, _ = two_view_scene(n_match, 0, 40, seed=100 + n_match, noise=0.1) if n_match else two_view_scene(0, 0, 40, seed=100)
    gpu_ctx.upload_image(0, dA)
    gpu_ctx.upload_image(1, dB)
    gpu_ctx.upload_keypoints(0, kA)
    gpu_ctx.upload_keypoints(1, kB)
    pairs = np.array([(0, 1)], np.int32)
    exp_off, exp_qt, exp_d, (raw_off, _) = expected_lists(gpu_ctx, host, pairs, [kA, kB])
    offs, qt, d = gpu_ctx.match_pairs_verified(pairs)
    assert np.array_equal(offs, exp_off) and np.array_equal(qt, exp_qt)
    n_raw = int(raw_off[1])
    if n_raw < 7:
        assert offs[1] == 0
    elif n_raw == 7:
        assert offs[1] == 7
    elif n_match >= 8 and n_raw == n_match:
        assert offs[1] >= n_match - 1     # clean geometry: (almost) everything is an inlier


def test_parameters_and_errors(gpu_ctx, host):
    dA, kA, dB, kB, _ = two_view_scene(200, 200, 50, seed=9)
    gpu_ctx.upload_image(0, dA)
    gpu_ctx.upload_image(1, dB)
    from monocularsfm_amd import _lib
    with pytest.raises(_lib.MsfmError):                      # no keypoints yet (upload_image resets them)
        gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32))
    gpu_ctx.upload_keypoints(0, kA)
    gpu_ctx.upload_keypoints(1, kB)
    a = gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32))
    b = gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32))
    assert np.array_equal(a[1], b[1])                          # deterministic
    loose = gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32), threshold=30.0)
    tight = gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32), threshold=0.05)
    assert loose[0][1] >= a[0][1] >= tight[0][1]
    few = gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32), max_iters=3, seed=7)
    assert few[0][1] <= a[0][1] + 5
    with pytest.raises(_lib.MsfmError):
        gpu_ctx.match_pairs_verified(np.array([(0, 1)], np.int32), confidence=1.5)
    with pytest.raises(_lib.MsfmError):
        gpu_ctx.upload_keypoints(0, kA[:10])                   # fewer keypoints than descriptor rows
