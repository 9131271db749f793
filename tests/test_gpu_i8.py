"""Byte stores on the integer matrix cores (monocularsfm_amd/csrc/msfm_sweep_i8.hip.h): the three routes -- i8 MFMA
prefilter, fp16 MFMA prefilter, brute-force exact kernel -- must return the same bits, and those must be the exact-integer
reference's (oracle/int_oracle.py: on 0..255 data the reference's fp32 arithmetic is exact integer arithmetic,
/root/reference/src/Feature/FeatureMatching.cpp:171-190 through cv::BFMatcher)."""
import numpy as np
import pytest

from monocularsfm_amd import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
I8, F16, BRUTE = 1, 2, 0


def b(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same_result(x, y):
    return np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(b(x[2]), b(y[2]))


def run_modes(ctx, pairs, **kw):
    out = {}
    try:
        for mode in (I8, F16, BRUTE):
            ctx.set_prefilter(mode)
            out[mode] = ctx.match_pairs(pairs, **kw)
            out[mode, "i8"] = ctx.profile()["sweep1_i8_launches"]   # (the profile describes the last call)
    finally:
        ctx.set_prefilter(True)
    return out


def check_vs_int_reference(u, pairs, res, **kw):
    from oracle import int_oracle as io
    offs, qt, d = res
    total = 0
    for p, (i, j) in enumerate(pairs):
        q, t, dd = io.match_pair(u[i], u[j], **kw)
        s, e = offs[p], offs[p + 1]
        assert np.array_equal(qt[s:e, 0], q) and np.array_equal(qt[s:e, 1], t) and np.array_equal(b(d[s:e]), b(dd)), (i, j, kw)
        total += len(q)
    return total


def test_three_routes_agree_with_the_integer_reference(gpu_ctx):
    """Ragged sizes (below one wave, one row, not a multiple of anything, an empty image), duplicates (ties), rows of all
    0 / all 255 (the extreme shifted norms 2^21 and 127^2 * 128), pairs of an image with itself excluded."""
    sizes = [700, 513, 64, 1, 1290, 33, 0, 2049]
    u = synth.u8_images(len(sizes), sizes, seed=808, dup_frac=0.15, as_float=False)
    u[0][5] = 0
    u[0][6] = 255
    u[1][9] = 0
    u[1][10] = 255
    u[4][100] = u[0][7]
    u[4][101] = u[0][7]          # two identical train rows: a distance tie for query 7 of image 0
    pairs = synth.all_pairs(len(sizes))
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)
    for kw in ({"ratio": 0.8, "cross_check": True, "max_distance": 1e9},
               {"ratio": 0.9, "cross_check": False, "max_distance": 500.0},
               {"ratio": 0.6, "cross_check": True, "max_distance": 300.0}):
        r = run_modes(gpu_ctx, pairs, **kw)
        assert r[I8, "i8"] >= 1 and r[F16, "i8"] == 0 and r[BRUTE, "i8"] == 0
        assert same_result(r[I8], r[F16]) and same_result(r[I8], r[BRUTE]), kw
        n = check_vs_int_reference(u, pairs, r[I8], **kw)
        assert n > 100, "test data: the pairs should have matches"
    gpu_ctx.clear_images()


def test_candidate_sets_are_tighter_on_the_integer_cores(gpu_ctx):
    """eps = 2 instead of ~1.5e-3 (n_a + n_b): the exact re-check sees fewer candidates, same result."""
    u = synth.u8_images(4, 3000, seed=31, as_float=False)
    pairs = synth.all_pairs(4)
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)
    cands = {}
    try:
        res = {}
        for mode in (I8, F16):
            gpu_ctx.set_prefilter(mode)
            res[mode] = gpu_ctx.match_pairs(pairs, ratio=0.8, cross_check=True, max_distance=1e9)
            cands[mode] = gpu_ctx.profile()["candidates"]
    finally:
        gpu_ctx.set_prefilter(True)
    assert same_result(res[I8], res[F16])
    assert 0 < cands[I8] <= cands[F16], cands
    gpu_ctx.clear_images()


def test_mixed_batch_takes_the_fp16_kernels(gpu_ctx, oracle):
    """A batch that joins a byte image with a float image (half-integers here: exact in fp32, not bytes) cannot use the
    integer cores.  The result does not depend on it."""
    u = synth.u8_images(3, [800, 900, 700], seed=5, as_float=False)
    imgs = [u[0].astype(F32), u[1].astype(F32) + F32(0.5), u[2].astype(F32)]
    gpu_ctx.upload_image(0, u[0])
    gpu_ctx.upload_image(1, imgs[1])
    gpu_ctx.upload_image(2, u[2])
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
    pairs = synth.all_pairs(3)
    mixed = gpu_ctx.match_pairs(pairs, **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 0 and gpu_ctx.profile()["prefilter_pairs"] == len(pairs)
    # ... but it is counted: pair (2, 0) joins two byte images and lost the integer route to its sub-batch's company
    assert gpu_ctx.profile()["demoted_pairs"] == 1
    o_offs, oq, ot, od = oracle.match_pairs(imgs, pairs, nthreads=8, **kw)
    assert np.array_equal(mixed[0], o_offs) and np.array_equal(mixed[1][:, 0], oq) and np.array_equal(mixed[1][:, 1], ot)
    assert np.array_equal(b(mixed[2]), b(od)) and o_offs[-1] > 50
    only_bytes = gpu_ctx.match_pairs(np.array([[2, 0]], np.int32), **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 1 and gpu_ctx.profile()["demoted_pairs"] == 0
    check_vs_int_reference(u, [(2, 0)], only_bytes, **kw)
    gpu_ctx.clear_images()


def test_float_upload_of_byte_values_is_a_byte_store(gpu_ctx, monkeypatch):
    """The reference's store is CV_32F throughout (Database::WriteDescriptors, /root/reference/src/Database/Database.cpp:249-262);
    raw OpenCV SIFT rows in it are integers 0..255.  Such an upload is recognised (every value checked on the device at
    upload) and served by the integer cores: same bits as the MSFM_DTYPE_U8 upload, the fp16 route and brute force; one
    non-integer value anywhere keeps the image a float image; MSFM_BYTE_DETECT=0 switches the recognition off."""
    from monocularsfm_amd import _lib
    sizes = [1100, 64, 513, 900, 1]
    u = synth.u8_images(len(sizes), sizes, seed=2718, dup_frac=0.2, as_float=False)
    pairs = synth.all_pairs(len(sizes))
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
    for i, x in enumerate(u):
        gpu_ctx.upload_image(i, x)
    ref = gpu_ctx.match_pairs(pairs, **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 1 and ref[0][-1] > 100
    for i, x in enumerate(u):
        gpu_ctx.upload_image(i, x.astype(F32))
    out = run_modes(gpu_ctx, pairs, **kw)
    assert out[I8, "i8"] == 1 and out[F16, "i8"] == 0 and out[BRUTE, "i8"] == 0
    for mode in (I8, F16, BRUTE):
        assert same_result(out[mode], ref), mode
    assert gpu_ctx.profile()["order_sensitive_rows"] == 0          # exact integers under every accumulation order
    check_vs_int_reference(u, pairs, out[I8], **kw)
    # subsets of a recognised image are recognised as well
    gpu_ctx.subset_image(0, 10, np.arange(0, 1100, 3, dtype=np.int32))
    gpu_ctx.subset_image(3, 11, np.arange(0, 900, 2, dtype=np.int32))
    sub = gpu_ctx.match_pairs(np.array([[10, 11]], np.int32), **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 1
    check_vs_int_reference({10: np.ascontiguousarray(u[0][::3]), 11: np.ascontiguousarray(u[3][::2])}, [(10, 11)], sub, **kw)
    # one value off the integer grid / outside [0, 255] / negative zero is fine: float image or not
    for value, is_bytes in ((17.5, False), (256.0, False), (-1.0, False), (-0.0, True)):
        x = u[0].astype(F32)
        x[1099, 127] = value
        gpu_ctx.upload_image(0, x)
        got = gpu_ctx.match_pairs(np.array([[0, 3]], np.int32), **kw)
        assert gpu_ctx.profile()["sweep1_i8_launches"] == (1 if is_bytes else 0), value
        gpu_ctx.set_prefilter(0)
        try:
            assert same_result(got, gpu_ctx.match_pairs(np.array([[0, 3]], np.int32), **kw)), value
        finally:
            gpu_ctx.set_prefilter(True)
    gpu_ctx.clear_images()
    monkeypatch.setenv("MSFM_BYTE_DETECT", "0")
    with _lib.Context(0) as ctx:
        for i, x in enumerate(u):
            ctx.upload_image(i, x.astype(F32))
        off = ctx.match_pairs(pairs, **kw)
        assert ctx.profile()["sweep1_i8_launches"] == 0 and same_result(off, ref)


def test_subsets_of_byte_images_stay_bytes(gpu_ctx):
    """msfm_subset_image (the top-scale subsets of the pre-emptive filter) of a byte image is a byte image."""
    from monocularsfm_amd.matcher import AUX0
    u = synth.u8_images(2, [1500, 1400], seed=77, dup_frac=0.2, as_float=False)
    rng = np.random.default_rng(3)
    rows = [np.sort(rng.choice(len(x), 600, replace=False)).astype(np.int32) for x in u]
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)
        gpu_ctx.subset_image(i, AUX0 + i, rows[i])
    a, bb = AUX0, AUX0 + 1
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
    res = gpu_ctx.match_pairs(np.array([[a, bb]], np.int32), **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 1
    check_vs_int_reference({a: u[0][rows[0]], bb: u[1][rows[1]]}, [(a, bb)], res, **kw)
    gpu_ctx.clear_images()


def test_full_size_byte_pair_on_all_routes(gpu_ctx):
    """16384 x 16384 (BASELINE config 5's shape): 32 work items per direction, several mask bits in the reverse plan."""
    u = synth.u8_images(2, 16384, seed=4096, as_float=False)
    for i, im in enumerate(u):
        gpu_ctx.upload_image(i, im)
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
    r = run_modes(gpu_ctx, np.array([[0, 1]], np.int32), **kw)
    assert r[I8, "i8"] == 1
    assert same_result(r[I8], r[F16]) and same_result(r[I8], r[BRUTE])
    assert len(r[I8][1]) > 500
    gpu_ctx.clear_images()


def test_golden_byte_fixture_through_the_integer_cores(gpu_ctx):
    """tests/golden/u8_int64_400x380.npz (made by tests/golden/make_golden.py: C oracle == exact int64 reference): the
    match lists of the byte upload on the default route, which must be the integer-core one."""
    import glob
    import os
    g = np.load(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "u8_int64_400x380.npz"))[0])
    gpu_ctx.upload_image(0, g["desc1"])
    gpu_ctx.upload_image(1, g["desc2"])
    md = float(g["max_distance"])
    for cc in (1, 0):
        q, t, d = gpu_ctx.match_pair(0, 1, 0.8, bool(cc), md)
        assert gpu_ctx.profile()["sweep1_i8_launches"] == 1
        assert np.array_equal(q, g["o0_cc%d_q" % cc]) and np.array_equal(t, g["o0_cc%d_t" % cc]) and np.array_equal(b(d), b(g["o0_cc%d_d" % cc]))
    gpu_ctx.clear_images()


def test_norm_spread_beyond_the_digit_range_takes_the_fp16_kernels(gpu_ctx):
    """The digit k-step carries H0 - h within [-504 064, 507 903]: an image holding an all-128 row (h = 0) next to an
    all-0 row (h = 2^20) cannot be centred, is stored as a float image and matched on the fp16 cores -- same lists."""
    u = synth.u8_images(2, [500, 450], seed=9, dup_frac=0.3, as_float=False)
    wide = u[0].copy()
    wide[3] = 128
    wide[4] = 0
    gpu_ctx.upload_image(0, wide)
    gpu_ctx.upload_image(1, u[1])
    gpu_ctx.upload_image(2, u[0])
    kw = {"ratio": 0.8, "cross_check": True, "max_distance": 1e9}
    res = gpu_ctx.match_pairs(np.array([[0, 1]], np.int32), **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 0 and gpu_ctx.profile()["prefilter_pairs"] == 1
    check_vs_int_reference({0: wide, 1: u[1]}, [(0, 1)], res, **kw)
    res = gpu_ctx.match_pairs(np.array([[2, 1]], np.int32), **kw)
    assert gpu_ctx.profile()["sweep1_i8_launches"] == 1
    check_vs_int_reference({2: u[0], 1: u[1]}, [(2, 1)], res, **kw)
    gpu_ctx.clear_images()
