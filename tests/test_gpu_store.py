"""The descriptor store (csrc/msfm_store.hip.h, msfm_store_host.hip.h): uploads only copy, msfm_finalize_store / the first matching
call builds; byte images keep 184 B per row and derive their float forms on demand; results never depend on when something is built."""
import numpy as np
import pytest

from monocularsfm_amd import _lib, synth

pytestmark = pytest.mark.gpu
F32 = np.float32


def bits(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same(x, y):
    return all(np.array_equal(bits(a), bits(b)) for a, b in zip(x, y))


def test_uploads_wait_for_finalize_and_byte_stores_stay_lean(oracle):
    u = synth.u8_images(6, 4096, seed=91, dup_frac=0.2, as_float=False)
    pairs = synth.all_pairs(6)
    with _lib.Context(0) as ctx:
        for i, im in enumerate(u):
            ctx.upload_image(i, im)
        info = ctx.store_info()
        assert info["pending_images"] == 6 and info["rows"] == 6 * 4096 and ctx.image_rows(3) == 4096
        ctx.finalize_store()
        info = ctx.store_info()
        assert info["pending_images"] == 0
        assert info["device_bytes"] <= 0.4 * 1024 * info["rows"], info        # <= 0.4 KB per row (round 4: 2.05 KB)
        lean = info["device_bytes"]
        got = ctx.match_pairs(pairs, max_distance=1e9)
        assert ctx.profile()["sweep1_i8_launches"] >= 1 and ctx.store_info()["device_bytes"] == lean   # the integer route reads nothing else
        for p in (0, 7, 14):
            i, j = pairs[p]
            oq, ot, od = oracle.match_pair(u[i].astype(F32), u[j].astype(F32), 0.8, True, 1e9, nthreads=8)
            s, e = got[0][p], got[0][p + 1]
            assert np.array_equal(got[1][s:e, 0], oq) and np.array_equal(got[1][s:e, 1], ot) and np.array_equal(bits(got[2][s:e]), bits(od))
        # the kNN-level API and the fp16-only / brute-force routes need the float forms: derived now, same results
        k1 = ctx.knn2_pair(1, 0)
        assert ctx.store_info()["device_bytes"] > lean
        ctx.set_prefilter(2)
        f16 = ctx.match_pairs(pairs, max_distance=1e9)
        ctx.set_prefilter(False)
        brute = ctx.match_pairs(pairs[:4], max_distance=1e9)
        k0 = ctx.knn2_pair(1, 0)
        ctx.set_prefilter(True)
        assert same(got, f16) and np.array_equal(brute[1], got[1][:brute[0][-1]])
        for a, b in zip(k1, k0):
            assert same(a, b)


def test_results_do_not_depend_on_when_images_are_built():
    imgs = synth.rootsift_images(5, [900, 0, 1400, 1100, 700], seed=92, n_proto=2500)
    u = synth.u8_images(2, [800, 1200], seed=93, as_float=False)
    pairs = np.array([(1, 0), (2, 0), (3, 2), (4, 3), (4, 0), (5, 2), (6, 5), (5, 6), (2, 1)], np.int32)
    results = []
    for mode in ("eager", "lazy", "reupload"):
        with _lib.Context(0) as ctx:
            for i, im in enumerate(imgs + u):
                ctx.upload_image(i, im)
                if mode == "eager":
                    ctx.finalize_store()
            if mode == "reupload":          # a pending image replaced before it was ever built; a built one replaced afterwards
                ctx.upload_image(2, imgs[3])
                ctx.upload_image(2, imgs[2])
                ctx.finalize_store()
                ctx.upload_image(0, imgs[0][:10])
                ctx.upload_image(0, imgs[0])
            results.append(ctx.match_pairs(pairs, max_distance=1e9))
    assert same(results[0], results[1]) and same(results[0], results[2]) and results[0][0][-1] > 50


def test_subset_of_a_pending_image_and_of_a_byte_image(oracle):
    imgs = synth.rootsift_images(2, [1500, 1300], seed=94, n_proto=2500)
    u = synth.u8_images(2, [1400, 1000], seed=95, dup_frac=0.3, as_float=False)
    rng = np.random.default_rng(5)
    with _lib.Context(0) as ctx:
        for k, (a, b) in enumerate(((imgs[0], imgs[1]), (u[0], u[1]))):
            ra, rb = rng.choice(len(a), 100 if k == 0 else 1100, replace=False), rng.choice(len(b), 120 if k == 0 else 800, replace=False)
            ctx.upload_image(0, a)
            ctx.upload_image(1, b)
            ctx.subset_image(0, 10, ra)          # the source is still pending: built first
            ctx.subset_image(1, 11, rb)
            assert ctx.store_info()["pending_images"] == 2
            q, t, d = ctx.match_pair(10, 11, 0.8, True, 1e9)
            oq, ot, od = oracle.match_pair(a[ra].astype(F32), b[rb].astype(F32), 0.8, True, 1e9, nthreads=4)
            assert np.array_equal(q, oq) and np.array_equal(t, ot) and np.array_equal(bits(d), bits(od))
            if k == 1:
                assert len(q) > 3


def test_bulk_upload_is_built_in_waves():
    # more than the inbox holds at once (256 MiB): the library builds what is waiting and carries on; results as for small stores
    imgs = synth.rootsift_images(72, 8000, seed=96, n_proto=12000)     # 72 x 4.1 MB = 295 MB
    with _lib.Context(0) as ctx:
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        assert 0 < ctx.store_info()["pending_images"] < 72
        pairs = np.array([(71, 0), (1, 70), (20, 21)], np.int32)
        got = ctx.match_pairs(pairs)
        with _lib.Context(0) as ref:
            for i in (0, 1, 20, 21, 70, 71):
                ref.upload_image(i, imgs[i])
            assert same(got, ref.match_pairs(pairs))
