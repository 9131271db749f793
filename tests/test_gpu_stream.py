"""The streaming form of the C ABI (msfm_match_pairs_begin / _next): one device sub-batch per call, nothing accumulates in the
library -- the reference streams by construction, one transaction per <= 100 pairs (src/Feature/FeatureMatching.cpp:13, 70-72,
118-139).  The chunks, concatenated, must be the lists msfm_match_pairs returns, bit for bit, under every cut."""
import numpy as np
import pytest

from monocularsfm_amd import _lib, synth

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def collect(ctx, pairs, **kw):
    offs = [np.zeros(1, np.int64)]
    qt, d, sens = [], [], []
    nxt = 0
    chunks = 0
    for ch in ctx.match_pairs_stream(pairs, **kw):
        assert ch["first"] == nxt and ch["n_pairs"] > 0      # in pair order, contiguous
        assert ch["offsets"][0] == 0 and len(ch["offsets"]) == ch["n_pairs"] + 1 and ch["offsets"][-1] == len(ch["qt"]) == len(ch["dist"])
        offs.append(offs[-1][-1] + ch["offsets"][1:])
        qt.append(ch["qt"])
        d.append(ch["dist"])
        sens.append(ch["sensitive"])
        nxt += ch["n_pairs"]
        chunks += 1
    assert nxt == len(pairs)
    return np.concatenate(offs), np.concatenate(qt) if qt else np.zeros((0, 2), np.int32), np.concatenate(d) if d else np.zeros(0, np.float32), \
        np.concatenate(sens) if sens else np.zeros(0, np.int32), chunks


@pytest.mark.parametrize("kind", ["rootsift", "u8"])
def test_stream_equals_one_call_under_forced_cuts(kind):
    n_img = 12
    if kind == "rootsift":
        imgs = synth.rootsift_images(n_img, [1500, 700, 2100, 64, 900, 1300, 0, 1800, 1000, 400, 1600, 1200], seed=77, n_proto=4000)
        kw = {}
    else:
        imgs = synth.u8_images(n_img, [1500, 700, 2100, 64, 900, 1300, 1, 1800, 1000, 400, 1600, 1200], seed=78, dup_frac=0.2, as_float=False)
        kw = {"max_distance": 1e9}
    pairs = synth.all_pairs(n_img)
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        ref = ctx.match_pairs(pairs, **kw)
        cert = ctx.order_certificate(len(pairs))
        for limit in (0, 7, 1):
            ctx.set_limits(max_pairs_per_batch=limit)
            offs, qt, d, sens, chunks = collect(ctx, pairs, **kw)
            assert np.array_equal(offs, ref[0]) and np.array_equal(qt, ref[1]) and np.array_equal(bits(d), bits(ref[2]))
            assert np.array_equal(sens, cert)
            if limit:
                assert chunks == -(-len(pairs) // limit)
        ctx.set_limits()
        # a series left open: the store must not change under it; the next matching call abandons it
        ctx.set_limits(max_pairs_per_batch=7)
        it = ctx.match_pairs_stream(pairs, **kw)
        next(it)
        with pytest.raises(_lib.MsfmError) as err:
            ctx.upload_image(0, imgs[0])
        assert err.value.code == _lib.E_STATE
        with pytest.raises(_lib.MsfmError):
            ctx.clear_images()
        ctx.set_limits()
        again = ctx.match_pairs(pairs, **kw)
        assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])
        with pytest.raises(_lib.MsfmError):
            next(it)          # its series is gone (state error, not stale data)
        # an empty pair list: no chunk
        assert list(ctx.match_pairs_stream(np.zeros((0, 2), np.int32))) == []


def test_stream_with_verification_equals_the_verified_call():
    imgs = synth.rootsift_images(5, [1200, 1100, 900, 1000, 1300], seed=79, n_proto=2500)
    pairs = synth.all_pairs(5)
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
            ctx.upload_keypoints(i, synth.keypoints(len(im), seed=100 + i))
        ref = ctx.match_pairs_verified(pairs)
        ctx.set_limits(max_pairs_per_batch=3)
        offs, qt, d, _, chunks = collect(ctx, pairs, verified=True)
        assert chunks == 4
        assert np.array_equal(offs, ref[0]) and np.array_equal(qt, ref[1]) and np.array_equal(bits(d), bits(ref[2]))


def test_stream_device_pointers_hold_the_same_lists():
    imgs = synth.u8_images(4, [2000, 1500, 1800, 900], seed=80, dup_frac=0.3, as_float=False)
    pairs = synth.all_pairs(4)
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        ctx.set_limits(max_pairs_per_batch=2)
        seen = 0
        for ch in ctx.match_pairs_stream(pairs, max_distance=1e9, copy=False):
            m = len(ch["qt"])
            if m == 0:
                continue
            assert np.array_equal(ctx.read_device(ch["d_qt"], (m, 2), np.int32), ch["qt"])
            assert np.array_equal(bits(ctx.read_device(ch["d_dist"], (m,), np.float32)), bits(ch["dist"]))
            seen += m
        assert seen > 100


def test_knn2_pair_and_an_early_exit_both_end_an_open_series():
    """(1) msfm_knn2_pair in the middle of a series abandons it like any other matching call (it ran on scratch set 0 with the
    series' sub-batches in flight and the next _next handed out foreign offsets): the following _next is a state error, the kNN lists are
    the ones a fresh context gives.  (2) A consumer that breaks out of the generator leaves no series open: msfm_match_pairs_end
    unlocks the store, uploads work again, and a later series still equals the one-call result."""
    imgs = synth.u8_images(8, [900, 1400, 700, 1100, 1300, 600, 1000, 1200], seed=81, dup_frac=0.2, as_float=False)
    pairs = synth.all_pairs(8)
    kw = {"max_distance": 1e9}
    with _lib.Context(0) as ref_ctx:
        for i, im in enumerate(imgs):
            ref_ctx.upload_image(i, im)
        knn_ref = ref_ctx.knn2_pair(3, 1)
        ref = ref_ctx.match_pairs(pairs, **kw)
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        ctx.set_limits(max_pairs_per_batch=3)
        it = ctx.match_pairs_stream(pairs, **kw)
        first = next(it)
        assert first["n_pairs"] == 3
        knn = ctx.knn2_pair(3, 1)                       # sub-batches 2 and 3 of the series are in flight here
        for a, b in zip(knn, knn_ref):
            assert np.array_equal(bits(a), bits(b))
        with pytest.raises(_lib.MsfmError) as err:
            next(it)
        assert err.value.code == _lib.E_STATE
        ctx.upload_image(0, imgs[0])                    # the store is unlocked
        # (2) early exit
        for k, ch in enumerate(ctx.match_pairs_stream(pairs, **kw)):
            if k == 1:
                break
        ctx.upload_image(1, imgs[1])                    # no E_STATE: the generator's finally ended the series
        assert ctx._L.msfm_match_pairs_end(ctx._h) == 0   # nothing open: a no-op
        it = ctx.match_pairs_stream(pairs, **kw)
        next(it)
        it.close()                                      # GeneratorExit takes the same path
        ctx.clear_images()
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        offs, qt, d, _, chunks = collect(ctx, pairs, **kw)
        assert chunks == -(-len(pairs) // 3)
        assert np.array_equal(offs, ref[0]) and np.array_equal(qt, ref[1]) and np.array_equal(bits(d), bits(ref[2]))
