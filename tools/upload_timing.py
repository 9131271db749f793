"""Wall clock of uploading the bench job's 128 images + msfm_finalize_store on fresh contexts, one / two cores for the host copy
(MSFM_UPLOAD_THREADS): python tools/upload_timing.py   -> profiles/r05_upload_timing.txt"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
imgs, pairs, _ = synth.job("south-building", 128)
for threads in ("2", "1", "2", "1"):
    os.environ["MSFM_UPLOAD_THREADS"] = threads
    ctx = _lib.Context(0)
    t0 = time.perf_counter()
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    t1 = time.perf_counter()
    ctx.finalize_store()
    t2 = time.perf_counter()
    print("threads %s: upload loop %.2f ms, final finalize %.2f ms, total %.2f ms" % (threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), flush=True)
    ctx.close()
