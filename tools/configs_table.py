"""The routes of the matcher on the BASELINE configs, one table (profiles/rNN_configs.txt):
   integer matrix cores (v_mfma_i32_32x32x32_i8: byte stores directly; float stores in [0, 1] through their byte twins + an
   fp16 sweep of the surviving rows = route Q) | fp16 matrix cores only | VALU brute force (exact order).
Every route returns the same bits (asserted here on each job).  Large configs are timed on a seeded subset with the full
per-image size.  Usage: python tools/configs_table.py [--quick] > gpurun_out/configs.txt"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import _lib, synth  # noqa: E402

MODES = [(1, "MFMA i8 "), (2, "MFMA f16"), (0, "VALU f32")]   # (1 on a float store: route Q when every image has a byte twin)


def run(ctx, pairs, mode, steps, **kw):
    """-> (seconds per job with the default pipeline, per-sweep kernel times of ONE unpipelined job, result).
    The sweep times come from an unpipelined run: with two sub-batches in flight a launch's event span includes its wait
    for the other stream's sweep."""
    ctx.set_prefilter(mode)
    ctx.match_pairs(pairs, fetch="view", **kw)   # warm-up (buffer growth, capacity hints)
    ctx.match_pairs(pairs, fetch="view", **kw)
    ctx.set_pipeline(1)
    ctx.match_pairs(pairs, fetch="view", **kw)
    ctx.match_pairs(pairs, fetch="view", **kw)
    solo = ctx.profile()
    ctx.set_pipeline(0)
    ctx.match_pairs(pairs, fetch="view", **kw)
    acc = {"approx_kernel_ms": 0.0, "sweep2_ms": 0.0, "candidates": 0, "dist_kernel_ms": 0.0, "sweep1_i8_launches": 0,
           "sweep1_q8_launches": 0, "prefilter_descriptor_pairs": 0, "sweep1b_ms": 0.0}
    t0 = time.perf_counter()
    for _ in range(steps):
        offs, qt, d = ctx.match_pairs(pairs, fetch="view", **kw)
        p = ctx.profile()
        for k in acc:
            acc[k] += p[k]
    dt = (time.perf_counter() - t0) / steps
    res = (np.array(offs), np.array(qt), np.array(d).view(np.int32))
    out = {k: v / steps for k, v in acc.items()}
    for k in ("approx_kernel_ms", "sweep2_ms", "sweep1b_ms", "dist_kernel_ms"):
        out[k] = solo[k]
    return dt, out, res


def job(name, imgs, pairs, steps, byte_store, **kw):
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    rows = np.array([len(x) for x in imgs], np.int64)
    total = int((rows[pairs[:, 0]] * rows[pairs[:, 1]]).sum())
    n_rows = float((rows[pairs[:, 0]] + rows[pairs[:, 1]]).sum())
    print("%s: %d images, %d pairs, %.3g descriptor pairs per job" % (name, len(imgs), len(pairs), total))
    ref = None
    for mode, label in MODES:
        dt, a, res = run(ctx, pairs, mode, steps if mode else 1, **kw)
        if mode == 1 and not byte_store:
            if not a["sweep1_i8_launches"]:
                print("    %s  -- (float store without byte twins -- values outside [0, 1]: the integer cores are not used)" % label)
                continue
            if a["sweep1_q8_launches"]:
                label = "route Q "
        if ref is None:
            ref = res
        same = all(np.array_equal(x, y) for x, y in zip(ref, res))
        if mode:
            ops = 256.0 * a["prefilter_descriptor_pairs"] / max(1e-9, a["approx_kernel_ms"] * 1e-3) / 1e12
            peak = 5000.0 if a["sweep1_i8_launches"] else 2500.0
            detail = "sweep 1 %7.2f ms (%.0f T%s/s = %.3f of the dense peak)%s | sweep 2 %6.2f ms | %.2f candidates per row" % (
                a["approx_kernel_ms"], ops, "OP" if a["sweep1_i8_launches"] else "FLOP", ops / peak,
                (" | fp16 sweep 1' %6.2f ms" % a["sweep1b_ms"]) if a["sweep1b_ms"] else "", a["sweep2_ms"], a["candidates"] / n_rows)
        else:
            detail = "exact kernel %8.2f ms (%.1f TFLOP/s of 384 unfusable flop per pair)" % (
                a["dist_kernel_ms"], 384.0 * total / max(1e-9, a["dist_kernel_ms"] * 1e-3) / 1e12)
        print("    %s  %9.2f ms per job  %.3e desc-pairs/s | %s | same bits as the first route: %s" % (label, dt * 1e3, total / dt, detail, same), flush=True)
        assert same
    ctx.close()


def main():
    quick = "--quick" in sys.argv
    imgs, pairs, name = synth.job("south-building", 32 if quick else 128, None, seed=1234)
    job("config 2 (" + name + ")", imgs, pairs, 5, False)
    n4 = 16 if quick else 64
    imgs, pairs, name = synth.job("synthetic-u8", n4, 8192, seed=1329)
    job("config 4 subset (" + name + ", %d of 1329 images)" % n4, imgs, pairs, 3, True, max_distance=1e9)
    n5 = 8 if quick else 24
    imgs, pairs, name = synth.job("synthetic-u8", n5, 16384, seed=4096)
    job("config 5 subset (" + name + ", %d of 4096 images)" % n5, imgs, pairs, 3, True, max_distance=1e9)
    # the same byte values uploaded as floats (the reference's CV_32F store holding raw SIFT): recognised at upload, same route
    imgs, pairs, name = synth.job("synthetic-u8", n4, 8192, seed=1329)
    job("config 4 subset, byte values stored as float32 (recognised as a byte store at upload)", [x.astype(np.float32) for x in imgs], pairs, 3, False, max_distance=1e9)


if __name__ == "__main__":
    main()
