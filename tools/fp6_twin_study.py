"""Offline study (NumPy, no GPU) -- VERDICT r04 item 5: a block-scaled fp6 / fp4 FIRST sweep for float stores?

v_mfma_scale_f32_32x32x64_f8f6f4 multiplies K = 64 fp6 (e2m3) or fp4 (e2m1) values per instruction at ~2.3 x the rate of
v_mfma_i32_32x32x32_i8 (guide: 7.3-9.1 PF measured against >= 3.94 POP/s; tools/ubench_fp6.hip measures both on one box), with a
power-of-two scale per 32-element block of every operand row (E8M0) applied by the hardware.  Cascade under study:

    sweep 0   fp6 twins, every descriptor pair:   live / dead only (same triangle-inequality bound as route Q, msfm_q8.hip.h)
    sweep 1'  today's int8 twins, live rows only: thresholds for sweep 2 (compacted, both directions)
    sweep 2 + exact re-check as today.

A twin row is a^ = dequant(quant(a - mu)) + mu: the distance is shift-invariant, so a common shift mu (per store) is free and puts the
dense part of the value distribution where the e2m3 grid is fine (RootSIFT: most values near 0.05-0.1).  Per row the error norm
e_a = |a - a^|_2 is known at upload; | |a - b| - |a^ - b^| | <= e_a + e_b.  Row q is dead when
    L0 = sqrt(S^min) - (e_q + E) >= ratio * U1,  U1 = sqrt(S^(2)) + (e_q + E),   E = max error norm of the other image
(or L0 > max_distance).  This script measures, on seeded pairs of the bench generator (config 2):
  * error norms of int8 twins (today, s = 255 / 0.4375), fp6 e2m3 and fp4 e2m1 twins under several shifts / block-scale choices;
  * live-row fraction and candidates per live row of each;
and projects the step time of the cascade from measured kernel times (profiles/r04_bench_kernel_stats_pipeline1.txt).
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import synth  # noqa: E402


def grid(fmt):
    if fmt == "e2m3":   # fp6: 1 sign, 2 exponent (bias 1), 3 mantissa; no inf / nan
        mags = [m / 8.0 for m in range(8)]                                       # subnormals (e = 0): 0 .. 0.875
        mags += [(1 + m / 8.0) * 2.0 ** (e - 1) for e in (1, 2, 3) for m in range(8)]   # 1 .. 7.5
    elif fmt == "e3m2":  # bf6: bias 3
        mags = [m / 4.0 * 2.0 ** -2 for m in range(4)]
        mags += [(1 + m / 4.0) * 2.0 ** (e - 3) for e in range(1, 8) for m in range(4)]  # up to 28
    elif fmt == "e2m1":  # fp4: 0, .5, 1, 1.5, 2, 3, 4, 6
        mags = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]
    else:
        raise ValueError(fmt)
    g = np.array(sorted(set(mags)))
    return np.concatenate([-g[:0:-1], g])


def quant_block_scaled(x, fmt, mu, block=32, scale_mode="block"):
    """x [n, 128] float64 -> twin x^ (float64), with x - mu quantised on the fmt grid under a power-of-two scale per (row, 32-block)
    ("block"), per row ("row") or one for the whole image ("image")."""
    g = grid(fmt)
    gmax = g[-1]
    y = x - mu
    n = y.shape[0]
    yb = y.reshape(n, -1, block)
    if scale_mode == "block":
        amax = np.abs(yb).max(axis=2, keepdims=True)
    elif scale_mode == "row":
        amax = np.abs(yb).max(axis=(1, 2), keepdims=True)
    else:
        amax = np.full((1, 1, 1), np.abs(yb).max())
    amax = np.maximum(amax, 1e-30)
    k = np.ceil(np.log2(amax / gmax))          # E8M0: power of two, the block's largest value stays representable
    sc = 2.0 ** k
    z = yb / sc
    idx = np.clip(np.searchsorted(g, z), 1, len(g) - 1)
    lo, hi = g[idx - 1], g[idx]
    zq = np.where(z - lo <= hi - z, lo, hi)
    return (zq * sc).reshape(n, -1) + mu


def quant_int8(x, level):
    s = 255.0 / level
    q = np.clip(np.rint(x * s), 0, 255)
    return q / s


def bound_stats(S_hat, e_me, E_other, ratio, max_distance):
    """live mask and candidates per live row under the twins' bound (direct thresholds: T = U1^2)."""
    part = np.partition(S_hat, 1, axis=1)
    s0, s1 = np.maximum(part[:, 0], 0), np.maximum(part[:, 1], 0)
    err = e_me + E_other
    L0 = np.maximum(np.sqrt(s0) - err, 0)
    U1 = np.sqrt(s1) + err
    dead = (L0 >= ratio * U1) | (L0 > max_distance)
    return ~dead, U1


def study(n_images=24, n_pairs=10, seed=1234, ratio=0.8, max_distance=0.7):
    imgs, pairs, _ = synth.job("south-building", n_images, seed=seed)
    rng = np.random.default_rng(7)
    sel = rng.choice(len(pairs), n_pairs, replace=False)
    allv = np.concatenate([x.ravel() for x in imgs[:4]])
    print("value distribution of the store: median %.4f, mean %.4f, p90 %.4f, p99 %.4f, max %.4f" % (
        np.median(allv), allv.mean(), np.quantile(allv, 0.9), np.quantile(allv, 0.99), max(float(x.max()) for x in imgs)))
    med = float(np.median(allv))
    schemes = [("int8 twins today (s = 255 / 0.4375)", lambda x: quant_int8(x, 0.4375))]
    for fmt in ("e2m3", "e3m2", "e2m1"):
        for mu_name, mu in (("0", 0.0), ("median", med)):
            for mode in ("block", "row"):
                schemes.append(("%s shift %s scale per %s" % (fmt, mu_name, mode),
                                (lambda x, fmt=fmt, mu=mu, mode=mode: quant_block_scaled(x, fmt, mu, scale_mode=mode))))
    need = sorted(set(int(i) for i in pairs[sel].ravel()))
    X = {i: imgs[i].astype(np.float64) for i in need}
    results = []
    for name, fn in schemes:
        tw = {i: fn(X[i]) for i in need}
        err = {i: np.linalg.norm(X[i] - tw[i], axis=1) for i in need}
        rows = live = cand = 0
        for i, j in pairs[sel]:
            a, b = tw[int(i)], tw[int(j)]
            S_hat = np.maximum((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T, 0)
            for d in (0, 1):
                Sx = S_hat if d == 0 else S_hat.T
                e_me, e_ot = (err[int(i)], err[int(j)]) if d == 0 else (err[int(j)], err[int(i)])
                lv, U1 = bound_stats(Sx, e_me, float(e_ot.max()), ratio, max_distance)
                rows += len(lv)
                live += int(lv.sum())
                cand += int((Sx[lv] <= (U1[lv] ** 2)[:, None]).sum())
        e_all = np.concatenate([err[i] for i in need])
        results.append((name, float(e_all.mean()), float(e_all.max()), live / rows, cand / max(1, live)))
        print("  %-44s error norm mean %.4f max %.4f | live rows %.3f | candidates per live row %.1f" % results[-1])
    return results


def project(results):
    """Step-time projection of the cascade for the bench job, from the round-4 kernel times (pipeline off):
    int8 sweep 1 26.9 ms (10 MFMA per tile and wave: 8 data + 2 digit; matrix duty 0.63, the VALU epilogue co-issues),
    sweep 2 5.5 ms, exact re-check 2.75 ms, the rest 2.8 ms."""
    s1_i8, s2, ex, rest = 26.9, 5.5, 2.75, 2.8
    today = s1_i8 + s2 + ex + rest
    print("\nprojection for the bench job (kernel times with the pipeline off; today %.1f ms):" % today)
    f_today = results[0][3]
    for name, e_mean, e_max, lf, cpl in results[1:]:
        # sweep 0: the 8 data MFMA of 34 cycles become 4 of 32 (K = 64), the norm step stays 2 x 32: matrix time 0.56 of today's;
        # the measured marginal value of matrix time in this kernel (r04_i8_structural_experiments.txt: 2 of 10 MFMA = 10 % of the
        # time) gives the OPTIMISTIC factor below; the epilogue's VALU work and the barriers do not shrink
        s0 = s1_i8 * (1.0 - 0.5 * (1.0 - 0.56 * 1.0) / (1.0 - 0.8) * 0.10 / 0.5)
        # sweep 1': int8 twins on the live rows, both directions compacted (live fraction lf of the rows each way), at 0.85 of
        # sweep 1's efficiency (what the compacted sweep 2 reaches)
        s1p = s1_i8 * 2.0 * lf / 0.85 if lf > f_today * 1.02 else s1_i8 * 2.0 * lf / 0.85
        total = s0 + s1p + s2 + ex + rest
        print("  %-44s sweep 0 %.1f ms + int8 sweep 1' on %.1f %% live rows %.1f ms + sweep 2 / re-check / rest %.1f ms = %.1f ms (%+.1f)" % (
            name, s0, 100 * lf, s1p, s2 + ex + rest, total, total - today))


if __name__ == "__main__":
    r = study()
    project(r)
