"""What does the power limiter charge sweep 1 for its OPERAND ENCODING?  (VERDICT r05, next #4.)

The integer route feeds the matrix cores `x' = x - 128` (csrc/msfm_store.hip.h): SIFT-like data has its mass near 0, so x' sits at
-128 .. -90, every product is ~ +16 384 and every accumulator bit toggles.  The part is power-limited under this kernel
(profiles/r03_power_trace.txt) and the guide records +19 % for zero-filled operands on one binary.  This is a TIMING probe of
`sweep_i8_kernel<1>` on byte stores whose operand bytes are chosen through the uploaded VALUES -- no kernel patch:

    sift          |N(0, 48)| clipped to 0..255         -> x' = -128 .. -80      (today's encoding on SIFT-like data)
    recentred     the same rows + (128 - median)       -> x' centred on 0       (what a per-store offset c = median would feed)
    positive      the same rows + 128 (clipped)        -> x' = x: 0 .. 127      (small POSITIVE operands: what c = 0 would feed)
    sparse_0      128 except 4 random entries per row  -> x' = 0 in 97 %        (the guide's zero-filled case: the size of the prize)
    sparse_m128   0 except 4 random entries per row    -> x' = -128 in 97 %     (large magnitude, nothing toggles between operands)
    uniform       uniform 0..255                       -> x' uniform            (the high-toggle end)

Every call runs with max_distance = 0.001 (no pair of distinct integer rows can pass the distance cut: all rows are provably dead
after sweep 1, so the tail is empty and the call is sweep 1 + thresholds; a NEGATIVE max_distance switches the pruning off instead) and pipeline 1 (one launch per call).  Sweep 1's control flow does not depend on
the data (every descriptor pair is multiplied, the epilogue is v_max3): the launch does the same work in every variant.
Beside every variant: tools/power_sampler (socket power, shader clocks, PPT residency at 20 Hz).

    python tools/operand_toggle.py [--images 64] [--desc 8192] [--seconds 8] > gpurun_out/operand_toggle.txt
"""
import argparse
import csv
import io
import os
import subprocess
import sys
import time

import numpy as np

MAXD = 1e-3   # below every distance of distinct integer rows (>= 1): every row is provably dead after sweep 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monocularsfm_amd import _lib, synth  # noqa: E402


def variants(n_images, n_desc, seed):
    base = synth.u8_images(n_images, n_desc, seed=seed, as_float=False)
    med = int(np.median(np.concatenate([b[:256].ravel() for b in base])))
    rng = np.random.default_rng(seed + 1)

    def sparse(fill):
        """every row `fill` except four random positions with random values: 97 % of the operand bytes constant, rows still distinct
        (an all-equal image has every distance 0: every row keeps all its candidates and the pairs fall back to brute force)"""
        out = []
        for b in base:
            a = np.full_like(b, fill)
            cols = rng.integers(0, 128, (len(b), 4))
            a[np.arange(len(b))[:, None], cols] = rng.integers(0, 256, (len(b), 4), dtype=np.uint8)
            out.append(a)
        return out
    yield "sift", base, "x' = x - 128 on |N(0,48)|: median x' %d" % (med - 128)
    yield "recentred", [np.clip(b.astype(np.int32) + (128 - med), 0, 255).astype(np.uint8) for b in base], \
        "the same rows + %d: median x' 0 (values beyond 255 clipped: timing probe)" % (128 - med)
    yield "positive", [np.clip(b.astype(np.int32) + 128, 0, 255).astype(np.uint8) for b in base], \
        "the same rows + 128: x' = x, small POSITIVE operands, median x' +%d (clipped at 127)" % med
    yield "sparse_0", sparse(128), "x' = 0 in 124 of 128 positions (the guide's zero-filled case, rows kept distinct)"
    yield "sparse_m128", sparse(0), "x' = -128 in 124 of 128 positions (large magnitude, nothing toggles)"
    yield "uniform", [rng.integers(0, 256, b.shape, dtype=np.uint8) for b in base], "x' uniform -128 .. 127"


def sample_power(seconds):
    exe = os.path.join(ROOT, "tools", "power_sampler")
    if not os.path.exists(exe):
        subprocess.run(["gcc", "-O2", "-I/opt/rocm/include", os.path.join(ROOT, "tools", "power_sampler.c"), "-L/opt/rocm/lib",
                        "-lrocm_smi64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=False)
    if not os.path.exists(exe):
        return None
    return subprocess.Popen([exe, str(seconds), "20"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


def power_stats(proc, t_begin, t_end, t_spawn):
    """mean socket power / clocks / PPT residency over the samples taken while the calls ran (first and last 0.5 s dropped)"""
    if proc is None:
        return {}
    text = proc.communicate()[0]
    body = [l for l in text.splitlines() if l.strip() and not l.startswith("#")]
    rows = []
    for r in csv.DictReader(io.StringIO("\n".join(body))):
        try:
            rows.append({k: float(v) for k, v in r.items()})
        except (TypeError, ValueError):
            pass
    lo, hi = t_begin - t_spawn + 0.5, t_end - t_spawn - 0.5
    b = [r for r in rows if lo <= r["t_s"] <= hi]
    if len(b) < 4:
        return {"samples": len(b)}
    dacc = b[-1]["accumulation_counter"] - b[0]["accumulation_counter"]
    de = b[-1]["energy_acc"] - b[0]["energy_acc"]
    dt = b[-1]["t_s"] - b[0]["t_s"]
    return {"samples": len(b), "power_W": sum(r["socket_power_W"] for r in b) / len(b),
            "power_energy_acc_W": de * 15.259e-6 / dt if dt > 0 and de > 0 else float("nan"),
            "gfxclk_MHz": sum(r["gfxclk_mean_MHz"] for r in b) / len(b),
            "gfxclk_min_MHz": min(r["gfxclk_min_MHz"] for r in b), "gfxclk_max_MHz": max(r["gfxclk_max_MHz"] for r in b),
            "ppt_residency_pct": 100.0 * (b[-1]["ppt_residency_acc"] - b[0]["ppt_residency_acc"]) / dacc if dacc > 0 else float("nan"),
            "hotspot_C": max(r["temp_hotspot_C"] for r in b)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--desc", type=int, default=8192)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--seed", type=int, default=1329)
    ap.add_argument("--rounds", type=int, default=2, help="every variant is run this many times, in alternation (clock drift of the box)")
    args = ap.parse_args()
    pairs = synth.all_pairs(args.images)
    dp = float(len(pairs)) * args.desc * args.desc
    print("# operand-toggle probe of sweep_i8_kernel<1>: %d images x %d byte descriptors, %d pairs, %.3e descriptor pairs per call" % (
        args.images, args.desc, len(pairs), dp))
    print("# max_distance = 0.001 (every row dead after sweep 1: empty tail), pipeline 1 (one launch per call), %.0f s per variant and round" % args.seconds)
    ctxs = {}
    notes = {}
    order = []
    for name, imgs, note in variants(args.images, args.desc, args.seed):
        c = _lib.Context(0)
        c.set_pipeline(1)
        for i, im in enumerate(imgs):
            c.upload_image(i, im)
        c.finalize_store()
        c.match_pairs(pairs, max_distance=MAXD, fetch=False)     # warm: buffers, plan hints
        ctxs[name], notes[name] = c, note
        order.append(name)
    print("# device: %s" % ctxs[order[0]].device_info())
    results = {n: [] for n in order}
    for rnd in range(args.rounds):
        for name in order:
            c = ctxs[name]
            time.sleep(1.5)                                      # let the clocks / the limiter's average relax between variants
            t_spawn = time.perf_counter()
            proc = sample_power(args.seconds + 3.0)
            time.sleep(0.7)
            s1_ms = s1_n = 0
            calls = 0
            cand = fb = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.seconds:
                c.match_pairs(pairs, max_distance=MAXD, fetch=False)
                p = c.profile()
                s1_ms += p["approx_kernel_ms"]
                s1_n += p["approx_kernel_launches"]
                cand += p["candidates"]
                fb += p["fallback_pairs"]
                calls += 1
            t1 = time.perf_counter()
            st = power_stats(proc, t0, t1, t_spawn)
            st.update({"round": rnd, "calls": calls, "call_ms": (t1 - t0) * 1e3 / calls, "s1_ms_per_launch": s1_ms / max(1, s1_n),
                       "s1_launches_per_call": s1_n / calls, "candidates": cand, "fallback_pairs": fb,
                       "frac_of_5_POPs": 256.0 * dp * calls / (s1_ms * 1e-3) / 5e15 if s1_ms else 0.0})
            results[name].append(st)
    ref = np.mean([r["s1_ms_per_launch"] for r in results["sift"]])
    print("%-11s %-5s %12s %9s %8s %9s %9s %9s %7s  %s" % ("variant", "round", "sweep1 ms", "vs sift", "frac", "call ms", "power W", "clk MHz", "PPT %", "note"))
    for name in order:
        for r in results[name]:
            print("%-11s %-5d %12.3f %8.1f%% %8.3f %9.3f %9.0f %9.0f %7.1f  %s%s" % (
                name, r["round"], r["s1_ms_per_launch"], 100.0 * (ref / r["s1_ms_per_launch"] - 1.0), r["frac_of_5_POPs"], r["call_ms"],
                r.get("power_W", float("nan")), r.get("gfxclk_MHz", float("nan")), r.get("ppt_residency_pct", float("nan")), notes[name],
                "" if r["candidates"] == 0 and r["fallback_pairs"] == 0 else "  [tail NOT empty: %d candidates, %d fallback pairs]" % (r["candidates"], r["fallback_pairs"])))
    print("# 'vs sift' = sweep-1 launches per second relative to today's encoding on SIFT-like data (positive: faster).")
    for c in ctxs.values():
        c.close()


if __name__ == "__main__":
    main()
