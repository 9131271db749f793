import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
ctx = _lib.Context(0)
for order in (0, 1, 3):
    for (n1, n2) in [(5000, 4800), (1300, 700), (130, 4000), (64, 64), (2, 300)]:
        imgs = synth.rootsift_images(2, [n1, n2], seed=n1 + 3 * n2 + order, n_proto=max(n1, n2) * 2)
        ctx.set_accum_order(order)
        ctx.upload_image(0, imgs[0]); ctx.upload_image(1, imgs[1])
        for mode in (True, False):
            ctx.set_prefilter(mode)
            k = ctx.knn2_pair(0, 1)
            p = ctx.profile()
            print(order, n1, n2, mode, {x: p[x] for x in ("prefilter_pairs", "fallback_pairs", "dist_kernel_launches", "candidates", "plan_regrows", "tie_queue_regrows", "tie_rows")}, ctx.store_info(), flush=True)
        ctx.set_prefilter(True)
