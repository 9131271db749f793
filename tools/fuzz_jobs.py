"""Seeded fuzz of the JOB level: whole msfm_match_pairs calls on random stores under random cuts -- pairs per sub-batch, scratch
budget, pipeline parts -- against (1) the same call under the library's defaults, bit for bit (offsets, (q, t) rows, distance bits:
results do not depend on how a call is cut or overlapped), and (2) the C oracle on the whole pair list (small stores).
tools/fuzz_routes.py covers the routes pair by pair; this one covers what sits above them: sub-batch cuts by count and by memory,
the packed table uploads and batched fills, sets in flight, route decisions per sub-batch (mixed stores: both first sweeps in one sub-batch, or demoted pairs), re-runs.
Usage: python tools/fuzz_jobs.py [seed] [cases]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth
from oracle import c_oracle

F32 = np.float32


def bits(a):
    a = np.asarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same(x, y):
    return all(np.array_equal(bits(a), bits(b)) for a, b in zip(x, y))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    bad = 0
    n_sub = n_demoted = n_q8 = n_i8 = n_mixed = n_chunks = 0
    for case in range(cases):
        n_img = int(rng.integers(3, 20))
        big = rng.random() < 0.35
        menu = [1, 2, 5, 63, 64, 65, 130, 300, 513, 700, 1100] + ([1700, 2600, 4100] if big else [])
        sizes = [int(rng.choice(menu)) for _ in range(n_img)]
        kind = str(rng.choice(["rootsift", "rootsift", "bytes", "bytes", "mixed-store", "scaled"]))
        if kind == "rootsift":
            imgs = synth.rootsift_images(n_img, sizes, seed=7000 + case, n_proto=max(sizes) + 50, sigma=float(rng.choice([0.02, 0.05, 0.1])))
        elif kind == "bytes":
            imgs = synth.u8_images(n_img, sizes, seed=8000 + case, dup_frac=float(rng.choice([0.02, 0.1, 0.4])), as_float=False)
        elif kind == "mixed-store":   # twinned images, images beyond [0, 1], byte-valued floats: sub-batches that mix routes
            imgs = synth.rootsift_images(n_img, sizes, seed=9000 + case, n_proto=max(sizes) + 50)
            for i in range(n_img):
                r = rng.random()
                if r < 0.25:
                    imgs[i] = (imgs[i] * F32(3.0)).astype(F32)
                elif r < 0.4:
                    imgs[i] = synth.u8_images(1, [sizes[i]], seed=9500 + 31 * case + i, as_float=True)[0].astype(F32)
        else:
            imgs = [(x * F32(rng.choice([1e-3, 0.25, 7.0, 150.0]))).astype(F32) for x in synth.rootsift_images(n_img, sizes, seed=9900 + case, n_proto=max(sizes) + 50)]
        ratio = float(rng.choice([0.6, 0.8, 0.95, 1.0]))
        cc = bool(rng.integers(0, 2))
        md = float(rng.choice([150.0, 400.0, np.inf])) if kind == "bytes" else float(rng.choice([0.3, 0.7, 2.0, np.inf]))
        allp = np.array([(i, j) for i in range(n_img) for j in range(n_img) if i != j], np.int32)
        rng.shuffle(allp)
        pairs = allp[:int(rng.integers(1, len(allp) + 1))]
        if rng.random() < 0.3:
            pairs = np.concatenate([pairs, pairs[:3], np.array([[0, 0]], np.int32)])   # repeats and a self pair
        ctx.clear_images()
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)
        ctx.set_limits(0, 0)
        ctx.set_pipeline(0)
        ref = ctx.match_pairs(pairs, ratio, cc, md)
        pr = ctx.profile()
        n_q8 += pr["sweep1_q8_launches"]
        n_i8 += pr["sweep1_i8_launches"]
        n_demoted += pr["demoted_pairs"]
        n_mixed += pr["mixed_route_sub_batches"]
        for _ in range(2):
            mp = int(rng.choice([1, 2, 3, 17, 100, 0]))
            sb = int(rng.choice([1 << 20, 8 << 20, 64 << 20, 1 << 30, 0]))
            parts = int(rng.choice([1, 2, 3, 6, 0]))
            ctx.set_limits(mp, sb)
            ctx.set_pipeline(parts)
            got = ctx.match_pairs(pairs, ratio, cc, md)
            n_sub += ctx.profile()["sub_batches"]
            n_mixed += ctx.profile()["mixed_route_sub_batches"]
            if not same(ref, got):
                bad += 1
                print("MISMATCH cut", case, kind, sizes, "max_pairs", mp, "scratch", sb, "parts", parts, flush=True)
            if rng.random() < 0.5:   # the same cut through the STREAMING form: the chunks, concatenated, are the call's lists
                offs, qts, ds = [np.zeros(1, np.int64)], [], []
                for ch in ctx.match_pairs_stream(pairs, ratio, cc, md):
                    offs.append(offs[-1][-1] + ch["offsets"][1:])
                    qts.append(ch["qt"])
                    ds.append(ch["dist"])
                    n_chunks += 1
                st = (np.concatenate(offs), np.concatenate(qts) if qts else np.zeros((0, 2), np.int32), np.concatenate(ds) if ds else np.zeros(0, F32))
                if not same(ref, st):
                    bad += 1
                    print("MISMATCH stream", case, kind, sizes, "max_pairs", mp, "scratch", sb, "parts", parts, flush=True)
        # a store rebuilt in another order / with images re-uploaded gives the same lists (uploads only copy; finalize builds)
        if rng.random() < 0.25:
            ctx.clear_images()
            for i in reversed(range(n_img)):
                ctx.upload_image(i, imgs[i])
                if rng.random() < 0.3:
                    ctx.finalize_store()
            ctx.set_limits(0, 0)
            ctx.set_pipeline(0)
            if not same(ref, ctx.match_pairs(pairs, ratio, cc, md)):
                bad += 1
                print("MISMATCH store order", case, kind, sizes, flush=True)
        # the oracle on the whole list (row products bounded: the C oracle does ~1e9 descriptor pairs a second and core)
        work = float(sum(sizes[i] * sizes[j] for i, j in pairs))
        if work < 6e8:
            o = c_oracle.match_pairs({i: np.asarray(im, F32) for i, im in enumerate(imgs)}, pairs, ratio, cc, md)
            qt = np.stack([o[1], o[2]], 1) if len(o[1]) else np.zeros((0, 2), np.int32)
            if not (np.array_equal(o[0], ref[0]) and np.array_equal(qt, ref[1]) and np.array_equal(bits(o[3]), bits(ref[2]))):
                bad += 1
                print("MISMATCH oracle", case, kind, sizes, ratio, cc, md, flush=True)
                for k, (i, j) in enumerate(pairs):   # which pairs, and what their images look like (largest value: twins exist up to 1)
                    a0, a1, b0, b1 = int(o[0][k]), int(o[0][k + 1]), int(ref[0][k]), int(ref[0][k + 1])
                    if a1 - a0 != b1 - b0 or not np.array_equal(qt[a0:a1], ref[1][b0:b1]) or not np.array_equal(bits(o[3][a0:a1]), bits(ref[2][b0:b1])):
                        print("   pair %d = (%d, %d): rows %d x %d, largest values %.4g / %.4g, oracle %d matches, device %d" % (
                            k, i, j, sizes[i], sizes[j], float(np.max(imgs[i])), float(np.max(imgs[j])), a1 - a0, b1 - b0), flush=True)
                print("   profile of the default call:", {x: pr[x] for x in ("sub_batches", "sweep1_q8_launches", "sweep1_i8_launches", "mixed_route_sub_batches",
                                                                           "demoted_pairs", "fallback_pairs", "prefilter_pairs", "plan_regrows")}, flush=True)
    print("job cases done, mismatches:", bad, "| sub-batches under the random cuts:", n_sub, "| route Q / integer sweep-1 launches (defaults):", n_q8, "/", n_i8,
          "| sub-batches with BOTH first sweeps (twins + fp16):", n_mixed, "| pairs demoted from the integer route in mixed sub-batches:", n_demoted,
          "| chunks of the streaming form compared:", n_chunks)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
