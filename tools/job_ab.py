"""A/B of library limits on ONE context and one resident store: the same synthetic job, run once per setting, alternated.

    python tools/job_ab.py --images 400 --desc 8192 --scratch-gib 48,128,48,128 [--max-pairs 16384] [--pipeline 6]

Per run: wall seconds, descriptor pairs / s, sub-batches, summed sweep-1 / sweep-2 event spans, the checksum of the result
(equal across settings: results do not depend on the cut).  `--once` runs the first setting once (for kernel traces)."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monocularsfm_amd import _lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synthetic-u8")
    ap.add_argument("--images", type=int, default=400)
    ap.add_argument("--desc", type=int, default=8192)
    ap.add_argument("--seed", type=int, default=1329)
    ap.add_argument("--scratch-gib", default="48")
    ap.add_argument("--max-pairs", default="16384")
    ap.add_argument("--pipeline", default="6")
    ap.add_argument("--warm", type=int, default=1, help="untimed calls before the series")
    args = ap.parse_args()
    imgs, pairs, name = synth.job(args.workload, args.images, args.desc if args.desc > 0 else None, seed=args.seed)
    n_rows = np.array([len(x) for x in imgs], np.int64)
    total = int((n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).sum())
    kw = {"max_distance": 1e9} if args.workload == "synthetic-u8" else {}
    ctx = _lib.Context(0)
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    sc = [float(x) for x in args.scratch_gib.split(",")]
    mp = [int(x) for x in args.max_pairs.split(",")]
    pl = [int(x) for x in args.pipeline.split(",")]
    n = max(len(sc), len(mp), len(pl))
    print("# %s: %d pairs, %.4g descriptor pairs" % (name, len(pairs), total), flush=True)
    for k in range(-args.warm, n):
        s, m, p = sc[max(k, 0) % len(sc)], mp[max(k, 0) % len(mp)], pl[max(k, 0) % len(pl)]
        ctx.set_limits(m, int(s * 2**30))
        ctx.set_pipeline(p)
        t0 = time.perf_counter()
        offs, qt, _ = ctx.match_pairs(pairs, fetch="view", **kw)
        dt = time.perf_counter() - t0
        pr = ctx.profile()
        crc = zlib.crc32(np.ascontiguousarray(qt).tobytes(), zlib.crc32(np.ascontiguousarray(offs).tobytes()))
        print(json.dumps({"warm": k < 0, "scratch_gib": s, "max_pairs": m, "pipeline": p, "wall_s": round(dt, 4), "value": total / dt,
                          "sub_batches": pr["sub_batches"], "sweep1_ms": round(pr["approx_kernel_ms"], 2), "sweep2_ms": round(pr["sweep2_ms"], 2),
                          "sweep1_frac": 256.0 * pr["prefilter_descriptor_pairs"] / max(1e-9, pr["approx_kernel_ms"] * 1e-3) / 5e15,
                          "candidates": pr["candidates"], "matches": int(offs[-1]), "plan_regrows": pr["plan_regrows"], "crc": crc}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
