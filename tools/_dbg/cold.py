import os, sys, time, json, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, name = synth.job("synthetic-u8", 400, 8192, seed=1329)
n_rows = np.array([len(x) for x in imgs], np.int64)
total = int((n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).sum())
for gib in (64, 16, 64, 16, 4):
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctx.finalize_store()
    ctx.set_limits(0, int(gib * 2**30))
    for call in range(3):
        t0 = time.perf_counter()
        offs, qt, _ = ctx.match_pairs(pairs, fetch="view", max_distance=1e9)
        dt = time.perf_counter() - t0
        print("budget %d GiB call %d: %.3f s, %d sub-batches, mem %s" % (gib, call, dt, ctx.profile()["sub_batches"], {k: round(v / 2**30, 2) for k, v in ctx.memory_info().items() if k in ("scratch", "results_device", "page_locked_host")}), flush=True)
    ctx.close()
