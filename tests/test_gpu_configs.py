"""BASELINE.json configs 3-5 as parity cases (scaled image counts, full per-image sizes): the shapes the
bench does not run.  At these sizes the oracle only checks sampled rows; the rest is covered by
size-independent properties (symmetry, prefilter == brute force, batch == single pair)."""
import numpy as np
import pytest

from monocularsfm_amd import synth

pytestmark = pytest.mark.gpu
F32 = np.float32


def b(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def check_pair_sampled(ctx, oracle, A, B, n_sample=48, seed=0, max_distance=0.7):
    fwd, rev = ctx.knn2_pair(0, 1)
    rng = np.random.default_rng(seed)
    rows = rng.choice(len(A), min(n_sample, len(A)), replace=False)
    oi0, od0, _, od1 = oracle.knn2(A[rows], B, 0, 8)
    assert np.array_equal(fwd[0][rows], oi0) and np.array_equal(b(fwd[1][rows]), b(od0)) and np.array_equal(b(fwd[2][rows]), b(od1))
    cols = rng.choice(len(B), min(n_sample, len(B)), replace=False)
    pi0, pd0, _, pd1 = oracle.knn2(B[cols], A, 0, 8)
    assert np.array_equal(rev[0][cols], pi0) and np.array_equal(b(rev[1][cols]), b(pd0)) and np.array_equal(b(rev[2][cols]), b(pd1))
    # symmetry and path equivalence
    fwd2, rev2 = ctx.knn2_pair(1, 0)
    for k in range(3):
        assert np.array_equal(b(fwd2[k]), b(rev[k])) and np.array_equal(b(rev2[k]), b(fwd[k]))
    ctx.set_prefilter(False)
    try:
        fwd3, rev3 = ctx.knn2_pair(0, 1)
    finally:
        ctx.set_prefilter(True)
    for k in range(3):
        assert np.array_equal(b(fwd3[k]), b(fwd[k])) and np.array_equal(b(rev3[k]), b(rev[k]))
    return fwd, rev


def test_config3_person_hall_shape(gpu_ctx, oracle):
    """Person-Hall: up to 8024 float RootSIFT descriptors per image (SIFTextractor.num_features cap)."""
    sizes = [8024, 5117, 7000, 6400]
    imgs = synth.rootsift_images(len(sizes), sizes, seed=1235, n_proto=20000)
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    check_pair_sampled(gpu_ctx, oracle, imgs[0], imgs[1], seed=3)
    pairs = np.array([(i, j) for i in range(len(sizes)) for j in range(i)], np.int32)
    offs, qt, d = gpu_ctx.match_pairs(pairs)
    assert offs[-1] > 1000
    for p, (i, j) in enumerate(pairs[:2]):
        oq, ot, od = oracle.match_pair(imgs[i], imgs[j], nthreads=8)
        s, e = offs[p], offs[p + 1]
        assert np.array_equal(qt[s:e, 0], oq) and np.array_equal(qt[s:e, 1], ot) and np.array_equal(b(d[s:e]), b(od))


def test_config4_u8_8192(gpu_ctx, oracle):
    """Synthetic 8192 x 128 u8-valued descriptors (integer distances: identical under every OpenCV build)."""
    u = synth.u8_images(3, 8192, seed=1329, as_float=False)
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)   # uint8 upload path
    A, B = u[0].astype(F32), u[1].astype(F32)
    fwd, rev = check_pair_sampled(gpu_ctx, oracle, A, B, seed=4)
    # the distances are sqrtf of INTEGERS (below 2^22 here, where rint(d^2) recovers S): d == fl32(sqrt(rint(d^2))) bit for bit
    for dd in (fwd[1], fwd[2], rev[1], rev[2]):
        dd = dd[dd < 3.0e38]
        S = np.rint(dd.astype(np.float64) ** 2)
        assert S.max() < 2 ** 22 and np.array_equal(np.sqrt(S).astype(F32).view(np.int32), dd.view(np.int32))
    q, t, d = gpu_ctx.match_pair(0, 1, 0.8, True, 1e9)
    assert len(q) > 200            # the planted near-duplicates match
    nz = q != 0
    assert np.array_equal(rev[0][t[nz]], q[nz]) and np.array_equal(fwd[0][q], t)


def test_config5_u8_16384(gpu_ctx, oracle):
    """Synthetic 16384 descriptors per image (the MFMA-vs-LDS config): 2.7e8 descriptor pairs per image pair."""
    u = synth.u8_images(2, 16384, seed=4096, as_float=False)
    gpu_ctx.upload_image(0, u[0])
    gpu_ctx.upload_image(1, u[1])
    fwd, rev = check_pair_sampled(gpu_ctx, oracle, u[0].astype(F32), u[1].astype(F32), n_sample=24, seed=5)
    prof_pairs = 16384 * 16384
    q, t, d = gpu_ctx.match_pair(0, 1, 0.8, True, 1e9)
    p = gpu_ctx.profile()
    assert p["descriptor_pairs"] == prof_pairs and p["prefilter_pairs"] == 1
    assert len(q) > 400 and (np.diff(q) > 0).all()


def whole_result_properties(offs, qt, pairs, n_rows):
    """size-independent properties of a job's result (tools/config4_full.py checks the same on the full configs)"""
    assert (np.diff(offs) >= 0).all()
    pair_of = np.repeat(np.arange(len(pairs)), np.diff(offs))
    q, t = np.asarray(qt[:, 0]), np.asarray(qt[:, 1])
    assert ((q >= 0) & (q < n_rows[pairs[pair_of, 0]]) & (t >= 0) & (t < n_rows[pairs[pair_of, 1]])).all()
    same = pair_of[1:] == pair_of[:-1]
    assert (np.diff(q.astype(np.int64))[same] > 0).all()                                       # ascending queryIdx inside a pair
    assert (np.diff(np.sort(pair_of.astype(np.int64) * (1 << 20) + t)) != 0).all()            # cross-check: a train row at most once per pair


@pytest.mark.parametrize("images, desc, seed, scratch_mib", [(96, 8192, 1329, 4096), (40, 16384, 4096, 2048)])
def test_config4_and_config5_jobs_beyond_a_toy_subset(gpu_ctx, oracle, images, desc, seed, scratch_mib):
    """BASELINE configs[3] at 96 of its 1329 images (4560 pairs, 3.1e11 descriptor pairs) and configs[4] at 40 of its 4096 (780 pairs of
    16384-row images -- 32 row blocks per image: the edge of the 32-bit block mask --, 2.1e11 descriptor pairs), their scratch budgets
    lowered so that the calls are cut into >= 8 sub-batches, three in flight, as ONE msfm_match_pairs call each: the first and the last
    pair and seeded random ones against the C oracle (on byte values its sums are exact integers -- tests/test_int_oracle.py pins it
    to the int64 reference, which would take two minutes per pair at this size), the whole result through its size-independent
    properties.  The full configs: tools/config4_full.py -> profiles/r04_config4_full.json,
    r04_config5_512.json.  Replaces at this size: /root/reference/src/Feature/FeatureMatching.cpp:102-145."""
    imgs, pairs, _ = synth.job("synthetic-u8", images, desc, seed=seed)
    n_rows = np.array([len(x) for x in imgs], np.int64)
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    gpu_ctx.set_limits(0, scratch_mib << 20)
    try:
        offs, qt, d = gpu_ctx.match_pairs(pairs, max_distance=1e9, fetch="view")
        offs, qt, d = offs.copy(), qt.copy(), d.copy()
        p = gpu_ctx.profile()
    finally:
        gpu_ctx.set_limits(0, 0)
    assert p["sweep1_i8_launches"] == p["sub_batches"] and p["prefilter_pairs"] == len(pairs) and p["fallback_pairs"] == 0
    assert p["order_sensitive_rows"] == 0 and p["demoted_pairs"] == 0
    assert p["sub_batches"] >= 8
    assert offs[-1] > 300 * len(pairs)                     # the planted 5 % near-duplicates match
    whole_result_properties(offs, qt, pairs, n_rows)
    rng = np.random.default_rng(seed)
    sel = np.asarray(sorted(set([0, len(pairs) - 1] + rng.choice(len(pairs), 4, replace=False).tolist())), np.int64)
    f32 = {int(i): imgs[int(i)].astype(F32) for i in np.unique(pairs[sel])}
    o_offs, oq, ot, od = oracle.match_pairs(f32, pairs[sel], max_distance=1e9, nthreads=16)
    assert o_offs[-1] > 300 * len(sel)
    for k, pk in enumerate(sel):
        s, e = int(offs[pk]), int(offs[pk + 1])
        os_, oe = int(o_offs[k]), int(o_offs[k + 1])
        assert np.array_equal(qt[s:e, 0], oq[os_:oe]) and np.array_equal(qt[s:e, 1], ot[os_:oe]), pk
        assert np.array_equal(b(d[s:e]), b(od[os_:oe])), pk
    gpu_ctx.clear_images()


@pytest.mark.parametrize("byte_store", [False, True])
def test_very_tall_image_against_a_small_one(gpu_ctx, byte_store):
    """Edge of the size range: 140 005 rows (274 A blocks of 512 rows: the 32-bit block mask of the reverse plan covers
    9 blocks per bit) against 700 rows, both orientations, every matrix-core route against the brute-force one."""
    rng = np.random.default_rng(140005)
    if byte_store:
        small = synth.u8_images(1, 700, seed=3, dup_frac=0.0, as_float=False)[0]
        tall = synth.u8_images(1, 140005, seed=4, dup_frac=0.0, as_float=False)[0]
        rows = rng.choice(len(tall), 300, replace=False)
        tall[rows] = np.clip(small[:300].astype(np.int32) + rng.integers(-2, 3, (300, 128)), 0, 255).astype(np.uint8)
        kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
        modes = (1, 2, 0)
    else:
        small = synth.rootsift_images(1, 700, seed=3, n_proto=1500)[0]
        tall = synth.rootsift_images(1, 140005, seed=4, n_proto=200000)[0]
        rows = rng.choice(len(tall), 300, replace=False)
        v = np.abs(small[:300] * (1 + 0.03 * rng.standard_normal((300, 128)).astype(F32)))
        tall[rows] = v / np.linalg.norm(v, axis=1, keepdims=True)
        kw = {"ratio": 0.8, "cross_check": True, "max_distance": 0.7}
        modes = (1, 0)
    gpu_ctx.upload_image(0, tall)
    gpu_ctx.upload_image(1, small)
    pairs = np.array([[0, 1], [1, 0]], np.int32)
    res = {}
    try:
        for m in modes:
            gpu_ctx.set_prefilter(m)
            res[m] = gpu_ctx.match_pairs(pairs, **kw)
            if m:
                assert gpu_ctx.profile()["fallback_pairs"] == 0
    finally:
        gpu_ctx.set_prefilter(True)
    for m in modes[:-1]:
        for x, y in zip(res[m], res[0]):
            assert np.array_equal(b(x), b(y)), m
    offs = res[0][0]
    assert offs[1] >= 200 and offs[2] - offs[1] >= 200          # the planted rows match in both orientations
    gpu_ctx.clear_images()
