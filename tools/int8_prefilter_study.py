"""Offline study (NumPy, no GPU): would an INT8-QUANTISED first sweep pay for FLOAT stores?  (task 7.)

RootSIFT rows are bounded ([0, 1], unit norm).  Quantise every row of an image with the image's scale s = 127 / max|x|:
a^ = round(a s) / s.  With e_a = a - a^ (known exactly per row at upload):
        | |a^ - b^|^2 - |a - b|^2 |  <=  (|e_a| + |e_b|) (2 |a - b| + |e_a| + |e_b|)  =: eps(q, t)
and per row eps_row = (|e_a| + max|e_b|) (2 sqrt(S~max_of_interest) + ...) -- here evaluated with the rigorous per-row
form eps_row(q) = (|e_a[q]| + E_b) (2 (|a| + B) ) simplified for unit-norm data to (|e_a| + E_b) * 4 (an upper bound of
2 |a - b| + ... since |a - b| <= 2), AND with the sharper data-dependent form that uses the approximate distance
itself: |a - b| <= |a^ - b^| + |e_a| + |e_b|.  The sweep on v_mfma_i32_32x32x32_i8 runs at ~2.0 POP/s against
~1.24 PFLOP/s for fp16 (profiles/r02_configs.txt): sweep 1 of the bench job would take ~27 ms instead of ~43 ms -- IF the
weaker bound still prunes most rows, because every live row then needs a second, fp16-accurate pass:
        projected step = i8 sweep 1 (all pairs) + fp16 sweep 1' on the live rows + compacted sweep 2 + tail.
This script measures, on seeded pairs of the config-2 generator (bench.py's workload):
   * the live-row fraction under the pruning test of pf_thresholds_kernel with eps_fp16 (today) and eps_int8,
   * candidates per live row at T = S~(2) + 2 eps for both,
and prints the projected step time.  Build only if it is <= 42 ms (VERDICT's criterion)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from monocularsfm_amd import synth  # noqa: E402


def study(n_images=24, n_pairs=12, seed=1234, ratio=0.8, max_distance=0.7):
    imgs, pairs, _ = synth.job("south-building", n_images, seed=seed)
    rng = np.random.default_rng(7)
    sel = rng.choice(len(pairs), n_pairs, replace=False)
    q8, err = [], []
    for x in imgs:
        s = 127.0 / float(np.abs(x).max())
        q = np.rint(x.astype(np.float64) * s)
        q8.append((q, s))
        err.append(np.linalg.norm(x.astype(np.float64) - q / s, axis=1))
    tot = {k: 0.0 for k in ("rows", "live16", "live8s", "live8w", "cand16", "cand8s", "cand8w", "max_err", "max_eps_sharp")}
    for i, j in pairs[sel]:
        a, b = imgs[i].astype(np.float64), imgs[j].astype(np.float64)
        S = np.maximum((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T, 0)
        (qa, sa), (qb, sb) = q8[i], q8[j]
        ah, bh = qa / sa, qb / sb
        S8 = np.maximum((ah * ah).sum(1)[:, None] + (bh * bh).sum(1)[None, :] - 2 * ah @ bh.T, 0)   # what the integer cores give, exactly
        ea, eb = err[i], err[j]
        for direction in (0, 1):
            Sx, S8x = (S, S8) if direction == 0 else (S.T, S8.T)
            e_me, e_other = (ea, eb) if direction == 0 else (eb, ea)
            n = Sx.shape[0]
            E = e_other.max()
            # fp16 route today: eps = 1.5e-3 (na + nb) + ..., na = nb = 1
            eps16 = np.full(n, 1.5e-3 * 2 + 2e-6)
            # int8, worst-case form: |a - b| <= 2
            eps8w = (e_me + E) * (4.0 + e_me + E)
            part = np.partition(S8x, 1, axis=1)
            s0, s1 = part[:, 0], part[:, 1]
            # int8, sharp form on the two smallest: |a - b| <= sqrt(S~) + e  (monotone in S~: evaluate at the value in question)
            def eps8s_at(v):
                return (e_me + E) * (2.0 * (np.sqrt(v) + e_me + E) + e_me + E)
            p16 = np.partition(Sx, 1, axis=1)       # fp16 S~ differs from S by < eps16: use S itself as its stand-in
            for tag, (m0, m1), eps0, eps1 in (("16", (p16[:, 0], p16[:, 1]), eps16, eps16),
                                              ("8w", (s0, s1), eps8w, eps8w),
                                              ("8s", (s0, s1), eps8s_at(s0), eps8s_at(s1))):
                d0lb = np.sqrt(np.maximum(m0 - eps0, 0))
                d1ub = np.sqrt(m1 + eps1)
                dead = (d0lb >= ratio * d1ub) | (d0lb > max_distance)
                live = ~dead
                T = m1 + 2 * eps1
                src = Sx if tag == "16" else S8x
                cand = (src[live] <= T[live][:, None]).sum()
                tot["live" + tag] += live.sum()
                tot["cand" + tag] += cand
            tot["rows"] += n
        tot["max_err"] = max(tot["max_err"], float(np.abs(S8 - S).max()))
    r = tot["rows"]
    print("pairs %d (both directions: %d rows); max |S8 - S| observed %.4f; mean quantisation error norm %.4f" % (
        n_pairs, r, tot["max_err"], float(np.mean([e.mean() for e in err]))))
    out = {}
    for tag, name in (("16", "fp16 sweep (today)           "), ("8s", "int8 sweep, sharp per-row eps "), ("8w", "int8 sweep, worst-case eps    ")):
        lf = tot["live" + tag] / r
        cpl = tot["cand" + tag] / max(1.0, tot["live" + tag])
        out[tag] = (lf, cpl)
        print("  %s live rows %.3f, candidates per live row %.1f (per row %.2f)" % (name, lf, cpl, lf * cpl))
    # projection for the bench job (profiles/r02_step_timeline.txt): sweep 1 42.5 ms fp16; i8 rate 2.03 / 1.24 of it
    s1_f16, tail = 42.5, 10.3
    lf = out["8s"][0]
    s1_i8 = s1_f16 * 1.24 / 2.03
    proj = s1_i8 + lf * s1_f16 / 0.85 + tail
    print("projection (bench job): i8 sweep 1 %.1f ms + fp16 pass on the %.1f %% live rows %.1f ms + today's tail %.1f ms = %.1f ms "
          "(today %.1f ms; build criterion <= 42 ms)" % (s1_i8, 100 * lf, lf * s1_f16 / 0.85, tail, proj, s1_f16 + tail))
    return proj


if __name__ == "__main__":
    study()
