"""The device side of the multi-GPU exchange step, on one GPU.

The first 8-GPU run executes three things nothing else runs: msfm_fetch_matches_device (the lists copied device-to-device
into a caller-owned HBM tensor), ShardedMatcher.match_to_writer's device branch (send tensor filled by the library,
writer's device receive buffer + page-locked copy-out) and the all_reduce of the counts over RCCL.  Here they run with
ONE rank on the one GPU of the box: `bench.py --backend nccl --force-collectives`.

Reference: the independent pair loop these shard, /root/reference/src/Feature/FeatureMatching.cpp:14, and the 100-pair
flush of BruteFeatureMatcher::RunMatching (:102-145) whose rows the writer rank stores."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from monocularsfm_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def b(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


@pytest.fixture(scope="module")
def job():
    sizes = [900, 640, 1300, 257, 1024, 700, 130, 999, 512, 1100, 64, 801]
    imgs = synth.rootsift_images(len(sizes), sizes, seed=78, n_proto=2600)
    return imgs, synth.all_pairs(len(sizes))


def test_fetch_matches_device_equals_the_host_copy_and_the_oracle(gpu_ctx, oracle, job):
    import torch
    imgs, pairs = job
    for i, im in enumerate(imgs):
        gpu_ctx.upload_image(i, im)
        gpu_ctx.upload_keypoints(i, synth.keypoints(len(im), seed=500 + i))
    dev = torch.device("cuda", 0)
    try:
        for limits in ((0, 0), (7, 0), (1, 0)):          # one sub-batch, ten, one per pair: the device list accumulates
            gpu_ctx.set_limits(*limits)
            offs, _, _ = gpu_ctx.match_pairs(pairs, fetch=False)
            M = int(offs[-1])
            assert M > 1000
            qt_t = torch.full((M + 16, 2), -7, dtype=torch.int32, device=dev)   # 16 guard rows behind the payload
            d_t = torch.full((M + 16,), -7.0, dtype=torch.float32, device=dev)
            gpu_ctx.fetch_matches_device(qt_t.data_ptr(), d_t.data_ptr())
            h_offs, h_qt, h_d = gpu_ctx.match_pairs(pairs)                        # the host copy of the same job
            assert np.array_equal(offs, h_offs)
            assert np.array_equal(qt_t[:M].cpu().numpy(), h_qt), limits
            assert np.array_equal(b(d_t[:M].cpu().numpy()), b(h_d)), limits
            assert (qt_t[M:] == -7).all() and (d_t[M:] == -7.0).all()              # nothing written past the count
            # either pointer may be NULL
            qt2 = torch.zeros((M, 2), dtype=torch.int32, device=dev)
            gpu_ctx.match_pairs(pairs, fetch=False)
            gpu_ctx.fetch_matches_device(qt2.data_ptr(), None)
            assert np.array_equal(qt2.cpu().numpy(), h_qt)
        # against the oracle (every pair)
        o_offs, oq, ot, od = oracle.match_pairs(imgs, pairs, nthreads=8)
        assert np.array_equal(o_offs, h_offs) and np.array_equal(h_qt[:, 0], oq) and np.array_equal(h_qt[:, 1], ot)
        assert np.array_equal(b(h_d), b(od))
        # after a verified call the device lists are the verified ones
        gpu_ctx.set_limits(11, 0)
        v_offs, v_qt, v_d = gpu_ctx.match_pairs_verified(pairs)
        Mv = int(v_offs[-1])
        assert 0 < Mv <= M
        offs2, _, _ = gpu_ctx.match_pairs_verified(pairs, fetch=False)
        assert np.array_equal(offs2, v_offs)
        qt_v = torch.zeros((Mv, 2), dtype=torch.int32, device=dev)
        d_v = torch.zeros((Mv,), dtype=torch.float32, device=dev)
        gpu_ctx.fetch_matches_device(qt_v.data_ptr(), d_v.data_ptr())
        assert np.array_equal(qt_v.cpu().numpy(), v_qt) and np.array_equal(b(d_v.cpu().numpy()), b(v_d))
    finally:
        gpu_ctx.set_limits(0, 0)


def test_fetch_matches_device_without_results_is_a_state_error(built_lib):
    from monocularsfm_amd import _lib
    with _lib.Context(0) as ctx:
        with pytest.raises(_lib.MsfmError) as e:
            ctx.fetch_matches_device(0, 0)
        assert e.value.code == _lib.E_STATE


def _bench(*extra, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--images", "24", "--u8-images", "6", "--no-cpu-baseline", *extra],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_rccl_exchange_branch_with_one_rank():
    """all_reduce over RCCL + match_to_writer's device path (library -> HBM send tensor -> writer's device buffer ->
    page-locked host), one rank on this box's GPU: same lists as the plain single-rank step."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    plain = _bench()
    coll = _bench("--backend", "nccl", "--force-collectives", env=env)
    assert coll["config"]["matches_per_step"] == plain["config"]["matches_per_step"] > 0
    assert coll["config"]["image_pairs"] == 276
    assert coll["strong_u8"]["matches_per_step"] == plain["strong_u8"]["matches_per_step"] > 0
    assert coll["per_rank_ms"][0]["exchange"] > 0.0 and plain["per_rank_ms"][0]["exchange"] == 0.0
    assert coll["exchange_checksum"] == plain["exchange_checksum"]      # the (q, t) rows themselves, not only their number
