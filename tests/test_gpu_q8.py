"""Route Q: float stores whose values lie in [0, 1] (RootSIFT) get byte twins, their FIRST sweep runs on the integer matrix
cores (csrc/msfm_q8.hip.h).  Fine twins (values up to 0.625: scale >= 408) give the thresholds of sweep 2 directly; coarse
ones get an fp16 sweep 1' of the ~6 % of rows that survive first.  Either way it must return the same bits as the fp16
route, the brute-force route and the oracle -- the reference computes every pair with cv::BFMatcher
(/root/reference/src/Feature/FeatureUtils.cpp:141-174), there is no approximation to hide behind."""
import numpy as np
import pytest

from monocularsfm_amd import _lib, synth

pytestmark = pytest.mark.gpu
F32 = np.float32


def b(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same(x, y):
    return np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(b(x[2]), b(y[2]))


def check_vs_oracle(oracle, imgs, pairs, sel, res, **kw):
    offs, qt, d = res
    o_offs, oq, ot, od = oracle.match_pairs(imgs, pairs[sel], nthreads=8, **kw)
    for k, p in enumerate(sel):
        s, e = int(offs[p]), int(offs[p + 1])
        os_, oe = int(o_offs[k]), int(o_offs[k + 1])
        assert e - s == oe - os_, (p, e - s, oe - os_)
        assert np.array_equal(qt[s:e, 0], oq[os_:oe]) and np.array_equal(qt[s:e, 1], ot[os_:oe]), p
        assert np.array_equal(b(d[s:e]), b(od[os_:oe])), p
    return int(o_offs[-1])


def test_route_q_equals_the_fp16_route_brute_force_and_the_oracle(gpu_ctx, oracle, monkeypatch):
    imgs, pairs, _ = synth.job("south-building", 20, seed=77)         # 190 pairs of ~5000-row images: route Q by default
    gpu_ctx.clear_images()                                             # (the session's context: start from an empty store, level 0)
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    try:
        gpu_ctx.set_prefilter(1)
        q = gpu_ctx.match_pairs(pairs)
        pq = gpu_ctx.profile()
        # values up to ~0.42 -> level 0.4375, scale 583: thresholds straight from the twins' sweep, no sweep 1'
        assert pq["sweep1_q8_launches"] >= 1 and pq["sweep1b_launches"] == 0 and pq["sweep1_q8_launches"] == pq["sweep1_i8_launches"]
        assert pq["prefilter_pairs"] == len(pairs) and pq["fallback_pairs"] == 0
        assert 0 < pq["sweep2_descriptor_pairs"] < 0.4 * pq["prefilter_descriptor_pairs"]       # only survivors see fp16
        assert q[0][-1] > 20000
        gpu_ctx.set_prefilter(2)                                           # fp16 matrix cores for every image
        f = gpu_ctx.match_pairs(pairs)
        pf = gpu_ctx.profile()
        assert pf["sweep1_q8_launches"] == 0 and pf["sweep1_i8_launches"] == 0 and pf["prefilter_pairs"] == len(pairs)
        assert same(q, f)
        assert pf["order_sensitive_rows"] == pq["order_sensitive_rows"]
        gpu_ctx.set_prefilter(0)
        sub = np.arange(0, len(pairs), 7)
        e = gpu_ctx.match_pairs(pairs[sub])
        for k, p in enumerate(sub):
            assert np.array_equal(e[1][e[0][k]:e[0][k + 1]], q[1][q[0][p]:q[0][p + 1]])
            assert np.array_equal(b(e[2][e[0][k]:e[0][k + 1]]), b(q[2][q[0][p]:q[0][p + 1]]))
        # other parameters: a looser ratio, no cross-check, a tighter cut
        gpu_ctx.set_prefilter(1)
        kw = {"ratio": 0.9, "cross_check": False, "max_distance": 0.45}
        q2 = gpu_ctx.match_pairs(pairs, **kw)
        assert gpu_ctx.profile()["sweep1_q8_launches"] >= 1
        gpu_ctx.set_prefilter(2)
        assert same(q2, gpu_ctx.match_pairs(pairs, **kw))
        # forced sub-batches and pipeline depths: same lists
        gpu_ctx.set_prefilter(1)
        for limits, pipe in (((23, 0), 0), ((0, 0), 1), ((64, 0), 3)):
            gpu_ctx.set_limits(*limits)
            gpu_ctx.set_pipeline(pipe)
            assert same(q, gpu_ctx.match_pairs(pairs)), (limits, pipe)
    finally:
        gpu_ctx.set_prefilter(True)
        gpu_ctx.set_limits(0, 0)
        gpu_ctx.set_pipeline(0)
    sel = np.sort(np.random.default_rng(9).choice(len(pairs), 24, replace=False))
    assert check_vs_oracle(oracle, imgs, pairs, sel, q) > 24 * 50
    # the same job with the refinement sweep forced (what a store with values near 1 takes): same bits
    monkeypatch.setenv("MSFM_Q8_DIRECT", "0")
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        r = ctx.match_pairs(pairs)
        pr = ctx.profile()
        assert pr["sweep1_q8_launches"] >= 1 and pr["sweep1b_launches"] == pr["sweep1_q8_launches"]
        assert 0 < pr["sweep1b_descriptor_pairs"] < 0.4 * pr["prefilter_descriptor_pairs"]
        assert same(q, r)
        assert pr["candidates"] < pq["candidates"]             # (the refined thresholds are the tighter ones)


def test_twin_level_follows_the_store(gpu_ctx):
    """The context quantises with one scale 255 / m, m = the largest twinned value so far rounded up to 1/16.  An upload that
    raises m makes the older twins stale: they are rebuilt at the next matching call (and past m = 0.625 the batch goes
    through the refinement sweep).  Same bits throughout; msfm_clear_images starts over."""
    imgs = synth.rootsift_images(5, [2600, 2300, 2500, 2400, 2200], seed=15, n_proto=6000)
    pairs = synth.all_pairs(5)
    gpu_ctx.clear_images()
    try:
        gpu_ctx.set_prefilter(2)
        for i, im in enumerate(imgs):
            gpu_ctx.upload_image(i, im)
        ref = gpu_ctx.match_pairs(pairs)
        gpu_ctx.set_prefilter(1)
        got = gpu_ctx.match_pairs(pairs)
        p = gpu_ctx.profile()
        assert p["sweep1_q8_launches"] >= 1 and p["sweep1b_launches"] == 0 and same(got, ref)
        # image 4 again with one large value (a legal RootSIFT row: the rest of that row shrinks): level 0.9375
        big = imgs[4].copy()
        big[7] *= F32(0.35)
        big[7, 3] = F32(0.93)
        gpu_ctx.upload_image(4, big)
        gpu_ctx.set_prefilter(2)
        ref2 = gpu_ctx.match_pairs(pairs)
        gpu_ctx.set_prefilter(1)
        got2 = gpu_ctx.match_pairs(pairs)               # twins 0..3 are rebuilt at the new level here
        p = gpu_ctx.profile()
        assert p["sweep1_q8_launches"] >= 1 and p["sweep1b_launches"] == p["sweep1_q8_launches"] and same(got2, ref2)
        sub = pairs[:3]                                   # pairs of the old images only: still the store's (coarse) level
        got3 = gpu_ctx.match_pairs(sub)
        assert gpu_ctx.profile()["sweep1b_launches"] >= 1
        assert np.array_equal(got3[1], ref[1][:ref[0][3]])
        gpu_ctx.clear_images()
        for i, im in enumerate(imgs):
            gpu_ctx.upload_image(i, im)
        got4 = gpu_ctx.match_pairs(pairs)
        assert gpu_ctx.profile()["sweep1b_launches"] == 0 and same(got4, ref)
    finally:
        gpu_ctx.set_prefilter(True)
        gpu_ctx.clear_images()


def test_twins_only_for_values_in_the_unit_interval(gpu_ctx):
    imgs = synth.rootsift_images(4, [2600, 2300, 2500, 2400], seed=5, n_proto=6000)
    pairs = synth.all_pairs(4)
    try:
        gpu_ctx.set_prefilter(1)
        for i, im in enumerate(imgs):
            gpu_ctx.upload_image(i, im)
        ref = gpu_ctx.match_pairs(pairs)
        assert gpu_ctx.profile()["sweep1_q8_launches"] >= 1
        # one image with ONE value outside [0, 1]: no twin for it.  Its three pairs take the fp16 first sweep, the other three stay on
        # the integer cores (fine twins: a mixed sub-batch runs both sweeps; round 3 sent all six to the fp16 kernels)
        bad = imgs[2].copy()
        bad[0, int(np.argmin(bad[0]))] = -1e-3
        gpu_ctx.upload_image(2, bad)
        got = gpu_ctx.match_pairs(pairs)
        p = gpu_ctx.profile()
        assert p["prefilter_pairs"] == len(pairs) and p["fallback_pairs"] == 0
        assert p["sweep1_q8_launches"] == 1 and p["mixed_route_sub_batches"] == 1 and p["demoted_pairs"] == 0
        assert not same(got, ref)                                                     # (the edited row changed a list)
        gpu_ctx.set_prefilter(2)                                                      # every pair on the fp16 cores
        assert same(got, gpu_ctx.match_pairs(pairs)) and gpu_ctx.profile()["mixed_route_sub_batches"] == 0
        gpu_ctx.set_prefilter(0)
        assert same(got, gpu_ctx.match_pairs(pairs))
        gpu_ctx.set_prefilter(1)
        # the same under forced sub-batches: (0,1) alone is all twins, (0,2) alone all fp16, the rest mixed or not as they fall
        gpu_ctx.set_limits(2, 0)
        assert same(got, gpu_ctx.match_pairs(pairs))
        gpu_ctx.set_limits(0, 0)
        # eight times the values: outside the unit interval -> fp16 route, and (exact scaling by a power of two) the same index
        # lists with eightfold distances
        gpu_ctx.set_prefilter(1)
        assert max(float(im.max()) for im in imgs) * 8.0 > 1.0
        for i, im in enumerate(imgs):
            gpu_ctx.upload_image(i, (8.0 * im).astype(F32))
        big = gpu_ctx.match_pairs(pairs, max_distance=5.6)
        assert gpu_ctx.profile()["sweep1_q8_launches"] == 0
        assert np.array_equal(big[0], ref[0]) and np.array_equal(big[1], ref[1])
        assert np.array_equal(b(big[2]), b((8.0 * ref[2]).astype(F32)))
    finally:
        gpu_ctx.set_prefilter(True)


@pytest.mark.parametrize("direct", ["1", "2"])
def test_route_q_on_small_ragged_and_degenerate_images(oracle, monkeypatch, direct):
    """MSFM_Q8=2 lifts the 'real images only' limit: images below one wave, one 512-row block, ragged sizes, near-duplicate
    swarms (tight thresholds), rows of exact zeros and ones (quantise without error; level 1: the refinement sweep unless
    MSFM_Q8_DIRECT=2 forces the direct thresholds onto the coarse twins)."""
    monkeypatch.setenv("MSFM_Q8", "2")
    monkeypatch.setenv("MSFM_Q8_DIRECT", direct)
    rng = np.random.default_rng(11)
    sizes = [3, 64, 65, 511, 512, 513, 700, 1300, 40, 2]
    imgs = synth.rootsift_images(len(sizes), sizes, seed=31, n_proto=900)
    # a swarm: 60 jittered copies of one row in image 6 and of the same row in image 7
    base = imgs[6][10]
    for im in (imgs[6], imgs[7]):
        rows = rng.choice(len(im), 60, replace=False)
        jit = np.abs(base[None, :] + 0.002 * rng.standard_normal((60, 128))).astype(F32)
        im[rows] = jit / np.linalg.norm(jit, axis=1, keepdims=True)
    # exact corners of the cube: 0 / 1 entries
    imgs[5][0] = 0.0
    imgs[5][0, 7] = 1.0
    imgs[4][3] = 0.0
    imgs[4][3, 7] = 1.0
    for im in imgs:
        np.clip(im, 0.0, 1.0, out=im)
    pairs = synth.all_pairs(len(sizes))
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        q = ctx.match_pairs(pairs)
        p = ctx.profile()
        assert p["sweep1_q8_launches"] >= 1 and (p["sweep1b_launches"] >= 1) == (direct == "1")
        ctx.set_prefilter(2)
        f = ctx.match_pairs(pairs)
        assert ctx.profile()["sweep1_q8_launches"] == 0
        assert same(q, f)
        ctx.set_prefilter(1)
        for kw in ({"ratio": 0.95, "cross_check": True, "max_distance": 10.0}, {"ratio": 0.5, "cross_check": False, "max_distance": 0.3}):
            a = ctx.match_pairs(pairs, **kw)
            assert ctx.profile()["sweep1_q8_launches"] >= 1
            ctx.set_prefilter(0)
            assert same(a, ctx.match_pairs(pairs, **kw)), kw
            ctx.set_prefilter(1)
    check_vs_oracle(oracle, imgs, pairs, np.arange(len(pairs)), q)
