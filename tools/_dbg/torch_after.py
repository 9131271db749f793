import os, sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
mode = sys.argv[1]
imgs = synth.rootsift_images(2, [900, 640], seed=78, n_proto=2600)
ctx = _lib.Context(0)
if mode != "noupload":
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
if mode == "match":
    ctx.match_pairs(np.array([[1, 0]], np.int32))
import torch
try:
    t = torch.zeros(4, device="cuda")
    print(mode, os.environ.get("MSFM_UPLOAD_THREADS"), "torch ok", flush=True)
except Exception as e:
    print(mode, os.environ.get("MSFM_UPLOAD_THREADS"), "torch FAILED", str(e)[:80], flush=True)
