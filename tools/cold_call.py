"""The first matching call of a context against the following ones: what a one-shot user (the CLI) pays that the bench's warm steps do not.
    [MSFM_DEBUG_TIMING=1] python tools/cold_call.py [south-building | u8] [images]
With MSFM_DEBUG_TIMING=1 the library reports, per call, what its allocations cost the host (hipMalloc / hipFree / page-locking) and every
buffer it re-grows.  Round 5 (profiles/r05_cold_call.txt): the 400-image byte job 0.83 s cold / 0.72 s warm before, 0.73 / 0.72 s after
the result lists stopped being page-locked in one block; 72 GiB of scratch cost 4 ms of hipMalloc."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "u8"
if kind == "u8":
    imgs, pairs, name = synth.job("synthetic-u8", int(sys.argv[2]) if len(sys.argv) > 2 else 400, 8192, seed=1329)
    kw = {"max_distance": 1e9}
else:
    imgs, pairs, name = synth.job("south-building", int(sys.argv[2]) if len(sys.argv) > 2 else 128)
    kw = {}
print("#", name, flush=True)
for rep in range(2):
    ctx = _lib.Context(0)
    t0 = time.perf_counter()
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctx.finalize_store()
    print("context %d: upload + build %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3), flush=True)
    for call in range(3):
        t0 = time.perf_counter()
        offs, qt, _ = ctx.match_pairs(pairs, fetch="view", **kw)
        dt = time.perf_counter() - t0
        p = ctx.profile()
        m = ctx.memory_info()
        print("context %d call %d: %.2f ms, %d sub-batches, plan_regrows %d | scratch %.2f GiB, result lists on the device %.2f GiB, page-locked %.2f GiB" % (
            rep, call, dt * 1e3, p["sub_batches"], p["plan_regrows"], m["scratch"] / 2**30, m["results_device"] / 2**30, m["page_locked_host"] / 2**30), flush=True)
    ctx.close()
