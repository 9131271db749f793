import os, sys, time, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, name = synth.job("south-building", 128)
for rep in range(2):
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctx.finalize_store()
    for call in range(3):
        t0 = time.perf_counter()
        offs, qt, _ = ctx.match_pairs(pairs, fetch="view")
        dt = time.perf_counter() - t0
        p = ctx.profile()
        print("rep %d call %d: %.2f ms, device %.2f ms, sub-batches %d, plan_regrows %d, fallback %d" % (rep, call, dt * 1e3, p["total_device_ms"], p["sub_batches"], p["plan_regrows"], p["fallback_pairs"]), flush=True)
    ctx.close()
