"""Long seeded fuzz: the prefilter routes (fp16 matrix cores; integer matrix cores for byte uploads; route Q = byte twins of
float images in [0, 1] -- run with MSFM_Q8=2 in the environment so that the small images of the fuzz take it too; fine twins
give sweep 2 its thresholds directly, coarse ones go through the fp16 sweep 1', MSFM_Q8_DIRECT=2 / =0 forces one of them) against
the brute-force route (match lists and knnMatch-level arrays) over random sizes, value types, scales, duplicates, NaNs,
parameters and orders.  Usage: [MSFM_Q8=2] python tools/fuzz_routes.py [seed] [cases]"""
import sys, numpy as np
sys.path.insert(0, '.')
from monocularsfm_amd import _lib, synth
F32 = np.float32
b = lambda a: np.asarray(a).view(np.int32) if np.asarray(a).dtype == np.float32 else np.asarray(a)
ctx = _lib.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
n_i8 = 0
n_q8 = 0
n_direct = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1500):
    n_img = int(rng.integers(2, 6))
    sizes = [int(rng.choice([1, 2, 3, 7, 31, 60, 64, 65, 130, 255, 256, 257, 600, 1100, 1700, 2600])) for _ in range(n_img)]
    kind = rng.choice(["rootsift", "u8", "bytes", "bytes", "gauss", "scaled", "mixed"])
    if kind == "rootsift":
        imgs = synth.rootsift_images(n_img, sizes, seed=1000 + case, n_proto=max(sizes) + 50, sigma=float(rng.choice([0.02, 0.05, 0.1])))
    elif kind == "u8":
        imgs = [x.astype(F32) for x in synth.u8_images(n_img, sizes, seed=2000 + case, as_float=True)]
    elif kind == "bytes":   # uint8 uploads: the integer-core route (extreme rows included)
        imgs = synth.u8_images(n_img, sizes, seed=5000 + case, dup_frac=float(rng.choice([0.02, 0.1, 0.4])), as_float=False)
        if rng.random() < 0.3:
            imgs[0][0] = 0
            imgs[-1][-1] = 255
        if rng.random() < 0.2:
            imgs[0] = (imgs[0] // 8).astype(np.uint8)   # a dark image: norms far from the other images'
    elif kind == "gauss":
        imgs = [rng.normal(size=(n, 128)).astype(F32) for n in sizes]
    elif kind == "scaled":
        sc = F32(rng.choice([1e-5, 1e-3, 0.25, 7.0, 150.0, 4000.0]))
        imgs = [(x * sc).astype(F32) for x in synth.rootsift_images(n_img, sizes, seed=3000 + case, n_proto=max(sizes) + 50)]
    else:
        imgs = [(x * F32(rng.choice([0.5, 1.0, 2.0, 4.0]))).astype(F32) for x in synth.rootsift_images(n_img, sizes, seed=4000 + case, n_proto=max(sizes) + 50)]
    if rng.random() < 0.5 and sizes[0] >= 3:
        imgs[0][1] = imgs[0][0]; imgs[-1][-1] = imgs[0][0]
    if rng.random() < 0.1 and imgs[0].dtype != np.uint8:
        imgs[0][0, 3] = np.nan
    order = int(rng.choice([0, 1, 3])); ratio = float(rng.choice([0.3, 0.6, 0.8, 0.95, 1.0, 1.2])); cc = bool(rng.integers(0, 2))
    md = float(rng.choice([0.05, 0.3, 0.7, 2.0, 1e4, np.inf]))
    if kind in ("u8", "bytes"):
        md = float(rng.choice([150.0, 400.0, 1e4, np.inf]))
    ctx.set_accum_order(order)
    ctx.clear_images()          # (the twins' level follows the store: every case starts from an empty one)
    for i, im in enumerate(imgs): ctx.upload_image(i, im)
    pairs = np.array([(i, j) for i in range(n_img) for j in range(n_img) if i != j or rng.random() < 0.2], np.int32)
    got = ctx.match_pairs(pairs, ratio, cc, md)
    n_i8 += ctx.profile()["sweep1_i8_launches"]
    n_q8 += ctx.profile()["sweep1_q8_launches"]
    n_direct += ctx.profile()["sweep1_q8_launches"] - ctx.profile()["sweep1b_launches"]
    if kind == "bytes" or ctx.profile()["sweep1_q8_launches"]:     # and the fp16 cores on the same data
        ctx.set_prefilter(2)
        got2 = ctx.match_pairs(pairs, ratio, cc, md)
        ctx.set_prefilter(True)
        if not (np.array_equal(got[0], got2[0]) and np.array_equal(got[1], got2[1]) and np.array_equal(b(got[2]), b(got2[2]))):
            bad += 1
            print("MISMATCH i8 vs f16", case, sizes, order, ratio, cc, md, flush=True)
    kp = [ctx.knn2_pair(int(i), int(j)) for i, j in pairs[:3]]
    ctx.set_prefilter(False)
    ref = ctx.match_pairs(pairs, ratio, cc, md)
    kb = [ctx.knn2_pair(int(i), int(j)) for i, j in pairs[:3]]
    ctx.set_prefilter(True)
    ok = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(b(got[2]), b(ref[2]))
    for x, y in zip(kp, kb):
        for d in (0, 1):
            for k in range(3):
                ok &= np.array_equal(b(x[d][k]), b(y[d][k]))
    if not ok:
        bad += 1
        print("MISMATCH", case, kind, sizes, order, ratio, cc, md, flush=True)
print("cases done, mismatches:", bad, "| integer-core sweep-1 launches:", n_i8, "| of them on byte twins of float images (route Q):", n_q8,
      "| of those with direct thresholds (no sweep 1'):", n_direct)
