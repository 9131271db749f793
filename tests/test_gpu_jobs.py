"""Job-level parity: the paths only a full-size batch reaches.

  (a) one msfm_match_pairs call forced into many device sub-batches (pair-count limit, scratch limit) ==
      the unsplit call == the oracle, on both routes and with the geometric verification hand-off;
  (b) the full BASELINE config-2 job -- 128 images x ~5000 descriptors, 8128 pairs in ONE launch per sweep (the
      cross-pair grouping of compacted live rows in sweep 2 only exists at this size) -- with 64 seeded pairs
      checked against the oracle and prefilter == brute force on a 500-pair subset;
  (c) config 3 at 330 images in one call (54 285 pairs => several sub-batches with the default limits), sampled
      pairs against the oracle;
  (d) integer descriptors against the exact-integer reference (oracle/int_oracle.py), u8 upload path.
The reference's counterpart of (a)-(c) is the pair loop of FeatureMatcher::MatchImagePairs and the 100-pair
flush of BruteFeatureMatcher::RunMatching (/root/reference/src/Feature/FeatureMatching.cpp:14-49, 118-139)."""
import os

import numpy as np
import pytest

from monocularsfm_amd import synth

pytestmark = pytest.mark.gpu
F32 = np.float32


def b(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same_result(x, y):
    return np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(b(x[2]), b(y[2]))


def check_pairs_vs_oracle(oracle, imgs, pairs, sel, offs, qt, d, nthreads=None, **kw):
    o_offs, oq, ot, od = oracle.match_pairs(imgs, pairs[sel], nthreads=nthreads, **kw)
    for k, p in enumerate(sel):
        s, e = int(offs[p]), int(offs[p + 1])
        os_, oe = int(o_offs[k]), int(o_offs[k + 1])
        assert e - s == oe - os_, (p, pairs[p], e - s, oe - os_)
        assert np.array_equal(qt[s:e, 0], oq[os_:oe]) and np.array_equal(qt[s:e, 1], ot[os_:oe]), (p, pairs[p])
        assert np.array_equal(b(d[s:e]), b(od[os_:oe])), (p, pairs[p])
    return int(o_offs[-1])


# ---- (a) forced sub-batches ------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def small_job():
    sizes = [900, 640, 1300, 257, 1024, 700, 130, 999, 512, 1100, 64, 801]
    imgs = synth.rootsift_images(len(sizes), sizes, seed=77, n_proto=2600)
    return imgs, synth.all_pairs(len(sizes))


@pytest.mark.parametrize("prefilter", [True, False])
def test_forced_sub_batches_equal_the_unsplit_call_and_the_oracle(gpu_ctx, oracle, small_job, prefilter):
    imgs, pairs = small_job
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    gpu_ctx.set_prefilter(prefilter)
    try:
        gpu_ctx.set_limits(0, 0)
        ref = gpu_ctx.match_pairs(pairs)
        assert gpu_ctx.profile()["sub_batches"] == 1
        assert ref[0][-1] > 1000
        for limits, min_batches in (((7, 0), 10), ((25, 0), 3), ((0, 4 << 20), 3), ((1, 0), len(pairs))):
            gpu_ctx.set_limits(*limits)
            got = gpu_ctx.match_pairs(pairs)
            nb = gpu_ctx.profile()["sub_batches"]
            assert nb >= min_batches, (limits, nb)
            assert same_result(ref, got), limits
            view = gpu_ctx.match_pairs(pairs, fetch="view")      # the page-locked buffers accumulate over sub-batches
            assert same_result(ref, view), limits
    finally:
        gpu_ctx.set_limits(0, 0)
        gpu_ctx.set_prefilter(True)
    total = check_pairs_vs_oracle(oracle, imgs, pairs, np.arange(len(pairs)), *ref, nthreads=8)
    assert total == ref[0][-1]


def test_forced_sub_batches_with_other_parameters_and_verification(gpu_ctx, small_job):
    imgs, pairs = small_job
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
        gpu_ctx.upload_keypoints(i, synth.keypoints(len(im), seed=900 + i))
    try:
        for kw in ({"ratio": 0.9, "cross_check": False, "max_distance": 0.5}, {"ratio": 1.2, "cross_check": True, "max_distance": 10.0}):
            gpu_ctx.set_limits(0, 0)
            ref = gpu_ctx.match_pairs(pairs, **kw)
            gpu_ctx.set_limits(9, 0)
            assert same_result(ref, gpu_ctx.match_pairs(pairs, **kw)), kw
        gpu_ctx.set_limits(0, 0)
        ref = gpu_ctx.match_pairs_verified(pairs)
        gpu_ctx.set_limits(11, 0)
        got = gpu_ctx.match_pairs_verified(pairs)
        assert gpu_ctx.profile()["sub_batches"] == 6
        assert same_result(ref, got)
    finally:
        gpu_ctx.set_limits(0, 0)


def test_limits_from_the_environment(gpu_ctx, small_job, monkeypatch):
    from monocularsfm_amd import _lib
    imgs, pairs = small_job
    monkeypatch.setenv("MSFM_MAX_PAIRS_PER_BATCH", "5")
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs[:6]):
            ctx.upload_image(i, im)
        ctx.match_pairs(pairs[:15])
        assert ctx.profile()["sub_batches"] == 3


# ---- (b) the full config-2 job ------------------------------------------------------------------------------

def test_config2_full_job_one_launch(gpu_ctx, oracle):
    imgs, pairs, _ = synth.job("south-building", 128, seed=1234)     # bench.py's N = 1 workload
    assert len(pairs) == 8128
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    gpu_ctx.set_limits(0, 0)
    gpu_ctx.set_pipeline(1)
    try:
        offs, qt, d = gpu_ctx.match_pairs(pairs)
        p = gpu_ctx.profile()
    finally:
        gpu_ctx.set_pipeline(0)
    assert p["sub_batches"] == 1 and p["approx_kernel_launches"] == 1      # ONE launch per sweep for the whole job
    # the default: the job cut into two equal sub-batches in flight (the first one's tail under the second one's sweep) == the same lists
    piped = gpu_ctx.match_pairs(pairs)
    pp_ = gpu_ctx.profile()
    assert pp_["sub_batches"] == 2 and pp_["approx_kernel_launches"] == 2 and pp_["prefilter_pairs"] == 8128
    # ... and round 3's schedule (six parts shrinking to 0.3 of the average, three in flight) as well
    gpu_ctx.set_pipeline(6)
    try:
        six = gpu_ctx.match_pairs(pairs)
        assert gpu_ctx.profile()["sub_batches"] == 6 and same_result((offs, qt, d), six)
    finally:
        gpu_ctx.set_pipeline(0)
    assert pp_["descriptor_pairs"] == p["descriptor_pairs"] and pp_["order_sensitive_rows"] == p["order_sensitive_rows"]
    assert same_result((offs, qt, d), piped)
    assert p["prefilter_pairs"] == 8128 and p["fallback_pairs"] == 0 and p["compacted_pairs"] > 7000
    n = np.array([len(x) for x in imgs], np.int64)
    assert p["descriptor_pairs"] == int((n[pairs[:, 0]] * n[pairs[:, 1]]).sum())
    # ascending queryIdx inside every pair, indices in range
    cnt = np.diff(offs)
    pid = np.repeat(np.arange(len(pairs)), cnt)
    assert (qt[:, 0] < n[pairs[pid, 0]]).all() and (qt[:, 1] < n[pairs[pid, 1]]).all() and (qt >= 0).all()
    inner = np.ones(len(qt), bool)
    inner[offs[:-1][cnt > 0]] = False
    assert (np.diff(qt[:, 0])[inner[1:]] > 0).all()
    assert (d <= F32(0.7)).all()
    # 64 seeded pairs against the oracle
    sel = np.sort(np.random.default_rng(2).choice(len(pairs), 64, replace=False))
    total = check_pairs_vs_oracle(oracle, imgs, pairs, sel, offs, qt, d)
    assert total > 64 * 50
    # prefilter == brute force on a 500-pair subset
    sub = np.sort(np.random.default_rng(3).choice(len(pairs), 500, replace=False))
    a = gpu_ctx.match_pairs(pairs[sub])
    gpu_ctx.set_prefilter(False)
    try:
        bres = gpu_ctx.match_pairs(pairs[sub])
        assert gpu_ctx.profile()["prefilter_pairs"] == 0
    finally:
        gpu_ctx.set_prefilter(True)
    assert same_result(a, bres)
    # and the subset call returns exactly the slices of the full-job call
    for k, pp in enumerate(sub):
        s, e = offs[pp], offs[pp + 1]
        assert np.array_equal(a[1][a[0][k]:a[0][k + 1]], qt[s:e])
    # the job in 3 sub-batches == the job in one
    gpu_ctx.set_limits(3000, 0)
    gpu_ctx.set_pipeline(1)
    try:
        split = gpu_ctx.match_pairs(pairs)
        assert gpu_ctx.profile()["sub_batches"] == 3
    finally:
        gpu_ctx.set_limits(0, 0)
        gpu_ctx.set_pipeline(0)
    assert same_result((offs, qt, d), split)


# ---- (c) config 3 ---------------------------------------------------------------------------------------------

def test_config3_330_images_in_one_call(gpu_ctx, oracle):
    imgs, pairs, _ = synth.job("south-building", 330, seed=1235)
    assert len(pairs) == 54285
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    gpu_ctx.set_limits(0, 0)
    offs, qt, d = gpu_ctx.match_pairs(pairs, fetch="view")
    offs, qt, d = offs.copy(), qt.copy(), d.copy()
    p = gpu_ctx.profile()
    assert p["sub_batches"] >= 4 and p["prefilter_pairs"] == 54285 and p["fallback_pairs"] == 0
    # sampled pairs, incl. the former sub-batch boundaries (multiples of 16384 pairs)
    rng = np.random.default_rng(5)
    edge = np.array([0, 16383, 16384, 32767, 32768, 49151, 49152, 54284])
    sel = np.unique(np.concatenate([edge, rng.choice(len(pairs), 24, replace=False)]))
    check_pairs_vs_oracle(oracle, imgs, pairs, sel, offs, qt, d)
    # a different cut of the same job gives the same lists
    gpu_ctx.set_limits(20000, 0)
    gpu_ctx.set_pipeline(1)
    try:
        again = gpu_ctx.match_pairs(pairs, fetch="view")
        assert gpu_ctx.profile()["sub_batches"] >= 3                          # (pair limit 20000, or the scratch limit of one set)
        assert same_result((offs, qt, d), again)
    finally:
        gpu_ctx.set_limits(0, 0)
        gpu_ctx.set_pipeline(0)
    gpu_ctx.clear_images()


@pytest.mark.parametrize("in_flight,parts,taper", [("1", "4", "1.0"), ("2", "4", "0.1"), ("3", "3", "1.0"), ("3", "8", "0.05")])
def test_results_do_not_depend_on_sets_in_flight_parts_or_taper(gpu_ctx, monkeypatch, in_flight, parts, taper):
    """The cost cut of a large call (msfm_set_pipeline / MSFM_PIPELINE parts, shrinking by MSFM_PIPELINE_TAPER) and the
    number of scratch sets in flight (MSFM_IN_FLIGHT; with three, sweep 1 of sub-batch k + 2 is ordered behind sweep 2 of k)
    are scheduling only: same offsets, same rows, same distance bits as the default context -- also with the pair limit
    forcing further cuts inside the parts, and on the fp16 route."""
    from monocularsfm_amd import _lib
    imgs, pairs, _ = synth.job("south-building", 72, seed=4321)        # 2556 pairs, 6.5e10 descriptor pairs: up to 4 parts (default: 2)
    gpu_ctx.clear_images()
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
    ref = gpu_ctx.match_pairs(pairs)
    p0 = gpu_ctx.profile()
    assert p0["sub_batches"] == 2 and p0["prefilter_pairs"] == len(pairs) and ref[0][-1] > 100000
    gpu_ctx.clear_images()
    monkeypatch.setenv("MSFM_IN_FLIGHT", in_flight)
    monkeypatch.setenv("MSFM_PIPELINE", parts)
    monkeypatch.setenv("MSFM_PIPELINE_TAPER", taper)
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        got = ctx.match_pairs(pairs)
        p = ctx.profile()
        assert same_result(ref, got)
        assert p["sub_batches"] == min(int(parts), 4) and p["order_sensitive_rows"] == p0["order_sensitive_rows"]
        ctx.set_limits(300, 0)                                         # the pair limit cuts inside the parts
        got = ctx.match_pairs(pairs)
        assert same_result(ref, got) and ctx.profile()["sub_batches"] >= 9
        ctx.set_limits(0, 0)
        ctx.set_prefilter(2)                                           # fp16 route, same schedule
        assert same_result(ref, ctx.match_pairs(pairs))


# ---- (d) integer descriptors vs the exact-integer reference -----------------------------------------------------

@pytest.mark.parametrize("prefilter", [True, False])
def test_u8_job_equals_the_integer_reference(gpu_ctx, prefilter):
    from oracle import int_oracle as io
    sizes = [1500, 1201, 640, 2048, 300]
    u = synth.u8_images(len(sizes), sizes, seed=4242, dup_frac=0.1, as_float=False)
    u[1][7] = u[1][3]
    u[0][10] = u[1][3]
    u[0][0] = u[1][40]
    pairs = synth.all_pairs(len(sizes))
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)      # uint8 upload
    gpu_ctx.set_prefilter(prefilter)
    try:
        for kw in ({"ratio": 0.8, "cross_check": True, "max_distance": 1e9}, {"ratio": 0.95, "cross_check": False, "max_distance": 420.0}):
            offs, qt, d = gpu_ctx.match_pairs(pairs, **kw)
            for p, (i, j) in enumerate(pairs):
                q, t, dd = io.match_pair(u[i], u[j], **kw)
                s, e = offs[p], offs[p + 1]
                assert np.array_equal(qt[s:e, 0], q) and np.array_equal(qt[s:e, 1], t) and np.array_equal(b(d[s:e]), b(dd)), (i, j, kw)
        fwd, rev = gpu_ctx.knn2_pair(0, 1)
        rf, rr = io.knn2(u[0], u[1]), io.knn2(u[1], u[0])
        assert np.array_equal(fwd[0], rf[0]) and np.array_equal(b(fwd[1]), b(rf[1])) and np.array_equal(b(fwd[2]), b(rf[3]))
        assert np.array_equal(rev[0], rr[0]) and np.array_equal(b(rev[1]), b(rr[1])) and np.array_equal(b(rev[2]), b(rr[3]))
    finally:
        gpu_ctx.set_prefilter(True)


@pytest.mark.parametrize("n_images,n_desc,seed", [(24, 8192, 1329), (10, 16384, 4096)])
def test_config4_and_5_byte_jobs_on_the_integer_cores(gpu_ctx, oracle, n_images, n_desc, seed):
    """BASELINE configs[3] / [4] at full per-image size, a seeded subset of the image set in ONE call: the cross-pair
    grouping of sweep 2 (several images share a streamed image, several mask bits per image at 16384 rows) on the
    integer-core route; sampled pairs against the C oracle, every pair against the fp16 route, a forced 3-way cut."""
    u, pairs, _ = synth.job("synthetic-u8", n_images, n_desc, seed=seed)
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
    offs, qt, d = gpu_ctx.match_pairs(pairs, **kw)
    p = gpu_ctx.profile()
    assert p["sweep1_i8_launches"] == 1 and p["sub_batches"] == 1 and p["fallback_pairs"] == 0 and p["prefilter_pairs"] == len(pairs)
    assert offs[-1] > 100 * len(pairs) // 4
    rng = np.random.default_rng(seed)
    sel = np.sort(rng.choice(len(pairs), 4, replace=False))
    check_pairs_vs_oracle(oracle, [x.astype(F32) for x in u], pairs, sel, offs, qt, d, **kw)
    try:
        gpu_ctx.set_prefilter(2)          # fp16 matrix cores on the same bytes
        assert same_result((offs, qt, d), gpu_ctx.match_pairs(pairs, **kw))
        assert gpu_ctx.profile()["sweep1_i8_launches"] == 0
        gpu_ctx.set_prefilter(True)
        gpu_ctx.set_limits((len(pairs) + 2) // 3, 0)
        assert same_result((offs, qt, d), gpu_ctx.match_pairs(pairs, **kw))
        assert gpu_ctx.profile()["sub_batches"] == 3 and gpu_ctx.profile()["sweep1_i8_launches"] == 3
    finally:
        gpu_ctx.set_prefilter(True)
        gpu_ctx.set_limits(0, 0)
    gpu_ctx.clear_images()


def test_tie_queue_grows_instead_of_failing(built_lib, oracle):
    """More sqrt-space ties in one batch than the initial queue holds (duplicate train descriptors; kNN-level API
    and ratio > 1 lists): the queue grows and the batch is re-run; round 1 returned MSFM_E_CAPACITY here."""
    from monocularsfm_amd import _lib
    rng = np.random.default_rng(11)
    base = synth.u8_images(1, 600, seed=99, as_float=True)[0]
    k = rng.integers(0, 40, 70000)
    A = np.clip(base[k] + np.rint(rng.normal(0, 3.0, (70000, 128))), 0, 255).astype(F32)
    B = np.concatenate([base[:40], base[:40], base[40:300]])      # train rows 0..39 == rows 40..79: every query ties
    rows = rng.choice(len(A), 300, replace=False)
    oi0, od0, _, od1 = oracle.knn2(A[rows], B, 0, 8)
    assert (od0 == od1).all() and (od0 > 0).all() and (oi0 < 40).all()
    oq, ot, od = oracle.match_pair(A[:2000], B, 1.5, False, 1e9, nthreads=8)
    assert len(oq) == 2000
    for mode in (True, False):
        with _lib.Context(0) as ctx:           # a fresh context: the queue starts at its initial capacity
            ctx.upload_image(0, A)
            ctx.upload_image(1, B)
            ctx.set_prefilter(mode)
            fwd, rev = ctx.knn2_pair(0, 1)
            prof = ctx.profile()
            assert prof["tie_rows"] >= 70000 and prof["tie_queue_regrows"] == 1, prof
            assert np.array_equal(fwd[0][rows], oi0) and np.array_equal(b(fwd[1][rows]), b(od0)) and np.array_equal(b(fwd[2][rows]), b(od1))
            assert (fwd[0] < 40).all() and (fwd[1] == fwd[2]).all()
            ctx.upload_image(2, A[:2000])
            q, t, d = ctx.match_pair(2, 1, 1.5, False, 1e9)     # ratio > 1: tied rows DO reach the list
            assert np.array_equal(q, oq) and np.array_equal(t, ot) and np.array_equal(b(d), b(od))
            q, t, d = ctx.match_pair(0, 1, 0.8, True, 1e9)       # ratio <= 1: no fix-up needed, none queued
            assert ctx.profile()["tie_queue_regrows"] == 0 and len(q) == 0


# ---- (e) the re-run path of the two-sub-batches-in-flight loop -----------------------------------------------------

@pytest.mark.parametrize("twins", [True, False])
def test_plan_regrow_with_sub_batches_in_flight(oracle, twins):
    """A fresh context sizes the sweep-2 plan from a guess (5/16 of the rows).  Data on which most rows stay alive
    (1150 rows of every image are a shared set, ratio 0.95) outgrows it: the sub-batch is dropped while its successor is
    already in flight on the other stream, everything is drained, it is re-run alone and the loop carries on behind it.
    Same lists as the brute-force route; the profile counts the re-runs."""
    from monocularsfm_amd import _lib
    rng = np.random.default_rng(3)
    sizes = [1500, 1400, 1600, 1300, 1550, 1450]
    imgs = synth.rootsift_images(len(sizes), sizes, seed=91, n_proto=4000)
    shared = imgs[0][:1150].copy()
    for k, im in enumerate(imgs):
        rows = rng.choice(len(im), 1150, replace=False)
        v = np.abs(shared * (1 + 0.01 * rng.standard_normal(shared.shape).astype(F32)))
        im[rows] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(F32)
        if not twins:
            im *= F32(4.0)          # values beyond 1: no byte twins, the fp16 route
    pairs = synth.all_pairs(len(sizes))
    kw = {"ratio": 0.95, "cross_check": True, "max_distance": 10.0}
    import os
    old = os.environ.get("MSFM_Q8")
    os.environ["MSFM_Q8"] = "2"
    try:
        with _lib.Context(0) as ctx:
            for i, im in enumerate(imgs):
                ctx.upload_image(i, im)
            ctx.set_limits(4, 0)                         # 15 pairs -> 4 sub-batches, three in flight
            got = ctx.match_pairs(pairs, **kw)
            p = ctx.profile()
            assert p["sub_batches"] == 4 and p["prefilter_pairs"] == len(pairs)
            assert p["plan_regrows"] >= 1, "the test data no longer outgrows the first plan"
            assert (p["sweep1_q8_launches"] > 0) == twins
            again = ctx.match_pairs(pairs, **kw)         # the hints fit now
            assert ctx.profile()["plan_regrows"] == 0 and same_result(got, again)
            ctx.set_prefilter(0)
            ctx.set_limits(0, 0)
            assert same_result(got, ctx.match_pairs(pairs, **kw))
    finally:
        if old is None:
            os.environ.pop("MSFM_Q8", None)
        else:
            os.environ["MSFM_Q8"] = old
    assert got[0][-1] > 3000
    check_pairs_vs_oracle(oracle, imgs, pairs, np.arange(len(pairs)), *got, nthreads=8, **kw)


@pytest.mark.gpu
def test_a_small_call_before_a_large_one_does_not_cost_the_large_one_a_re_run(oracle):
    """The ComputeMatches executable's order of calls in a fresh process: the pre-emptive filter (every pair on 100-row subsets:
    FeatureMatching.cpp:148-203) and then the pairs on the full images.  What the small call needed says nothing about the large one:
    the prediction of the sweep-2 plan's buffers is relative to the rows a sub-batch could compact at most and void beyond a factor
    two (round 5: the buffers kept from the small call made the large call's first sub-batch overflow its plan -- two sub-batches
    dropped and re-run, 45 ms of the executable's 0.41 s)."""
    from monocularsfm_amd import _lib
    n_img = 20
    imgs = synth.rootsift_images(n_img, [2400 + 37 * k for k in range(n_img)], seed=77, n_proto=6000)
    pairs = synth.all_pairs(n_img)
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 0.7}
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
            ctx.subset_image(i, 100 + i, np.arange(0, 100, dtype=np.int32) * 7)
        small = ctx.match_pairs(pairs + 100, **kw)
        assert ctx.profile()["prefilter_pairs"] == len(pairs)
        got = ctx.match_pairs(pairs, **kw)
        p = ctx.profile()
        assert p["prefilter_pairs"] == len(pairs) and p["plan_regrows"] == 0, p
        again = ctx.match_pairs(pairs, **kw)
        assert ctx.profile()["plan_regrows"] == 0 and same_result(got, again)
        small2 = ctx.match_pairs(pairs + 100, **kw)     # ... and back: the large call's buffers hold the small one
        assert ctx.profile()["plan_regrows"] == 0 and same_result(small, small2)
    assert got[0][-1] > 1000
    check_pairs_vs_oracle(oracle, imgs, pairs, np.arange(0, len(pairs), 9), *got, nthreads=8, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("q8", ["1", "2"])
def test_job_fuzz_random_stores_under_random_cuts(built_lib, q8):
    """tools/fuzz_jobs.py, a short run: whole calls on random stores (float / byte / mixed) under random cuts -- pairs per sub-batch,
    scratch budget, pipeline parts -- equal the call under the defaults bit for bit and the C oracle on the whole pair list.
    (3300 cases of it: profiles/r04_fuzz_jobs.txt.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSFM_Q8=q8)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_jobs.py"), "5", "30"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.parametrize("form", ["one_call", "stream"])
def test_out_of_device_memory_shrinks_the_sub_batches_and_carries_on(form):
    """The scratch budget of a call is derived from the device's free memory ONCE per state of the store (msfm_set_limits).  When another
    tenant of the GPU takes the memory afterwards -- here: a torch tensor that leaves ~0.7 GB -- a sub-batch no longer fits: the call
    gives its scratch back, halves the share of a sub-batch (at most a sixth of what is free) and cuts again from the same pair
    (msfm_profile.memory_shrinks), instead of failing with MSFM_E_DEVICE.  Same lists as on an empty device, from more sub-batches."""
    torch = pytest.importorskip("torch")
    from monocularsfm_amd import _lib
    imgs = synth.u8_images(40, 8192, seed=77, dup_frac=0.05, as_float=False)
    pairs = synth.all_pairs(40)                       # 780 pairs x ~1.9 MB of scratch
    small = synth.all_pairs(12)
    kw = {"max_distance": 1e9}
    with _lib.Context(0) as ref_ctx:
        for i, im in enumerate(imgs):
            ref_ctx.upload_image(i, im)
        ref = tuple(np.array(x) for x in ref_ctx.match_pairs(pairs, **kw))
        ref_batches = ref_ctx.profile()["sub_batches"]
    hog = None
    try:
        with _lib.Context(0) as ctx:
            for i, im in enumerate(imgs):
                ctx.upload_image(i, im)
            ctx.match_pairs(small, **kw)              # the budget of this store state is cached now: 64 GiB
            free = ctx.memory_info()["device_free"]
            hog = torch.empty(int(free - (700 << 20)), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            assert ctx.memory_info()["device_free"] < (1 << 30)
            if form == "one_call":
                got = ctx.match_pairs(pairs, **kw)
            else:
                offs, qt, d = [np.zeros(1, np.int64)], [], []
                for ch in ctx.match_pairs_stream(pairs, **kw):
                    offs.append(offs[-1][-1] + ch["offsets"][1:])
                    qt.append(ch["qt"])
                    d.append(ch["dist"])
                got = (np.concatenate(offs), np.concatenate(qt), np.concatenate(d))
            prof = ctx.profile()
            assert prof["memory_shrinks"] >= 1 and prof["sub_batches"] > ref_batches, prof
            assert same_result(got, ref)
            # the context remembers what worked: the next call on this store state starts from the smaller share
            again = ctx.match_pairs(pairs, **kw)
            assert same_result(again, ref) and ctx.profile()["memory_shrinks"] == 0
            # squeezed much harder (24 MB free, the budget derived afresh): the call either still fits what the context holds -- then the
            # lists are the same -- or fails with MSFM_E_DEVICE and a text that says memory; never a crash, and the context stays usable
            del hog
            hog = None
            torch.cuda.empty_cache()
            free = ctx.memory_info()["device_free"]
            hog = torch.empty(int(free - (24 << 20)), dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            ctx.set_limits()
            try:
                tight = ctx.match_pairs(pairs, **kw)
                assert same_result(tight, ref)
            except _lib.MsfmError as e:
                assert e.code == _lib.E_DEVICE and "memory" in str(e).lower()
            del hog
            hog = None
            torch.cuda.empty_cache()
            ok = ctx.match_pairs(small, **kw)         # and the context is usable afterwards
            assert len(ok[0]) == len(small) + 1
    finally:
        del hog
        torch.cuda.empty_cache()
