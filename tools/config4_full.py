"""A BASELINE byte config on one MI355X in ONE msfm_match_pairs call: configs[3] IN FULL (1329 images x 8192 byte descriptors,
all 882 456 pairs: the N = 1 anchor of north_star's >= 6x strong-scaling target) by default, or a seeded subset of configs[4]:

    python tools/config4_full.py > gpurun_out/config4_full.json
    python tools/config4_full.py --images 512 --desc 16384 --seed 4096 > gpurun_out/config5_512.json      (130 816 pairs)

`--stream`: the job through the STREAMING form of the C ABI (msfm_match_pairs_begin / _next): one device sub-batch per call, nothing
accumulates in the library -- the same checks on the chunks as they arrive (the lists of the sampled pairs are kept, the rest is
consumed and dropped), peak device / page-locked memory from msfm_memory_info after every chunk:

    python tools/config4_full.py --images 1024 --desc 16384 --seed 4096 --stream > gpurun_out/config5_1024_stream.json  (523 776 pairs)

`--warm`: one untimed call first (every buffer allocated, plan hints learnt): the timed call is then what a step of bench.py's
strong_u8 job measures; without it the call includes the first-touch cost of its scratch and result buffers (~10 ms per GiB).
Checks: (1) the call's sub-batch count against the count predicted from the library's scratch formula (msfm_pair_scratch_bytes,
csrc/msfm_hostutil.h; the cost marks of the pipeline may add up to five cuts), (2) sampled pairs -- the first and the last pair
of several sub-batches, plus seeded random ones -- against the C oracle, and some of them against the exact-integer reference
(oracle/int_oracle.py), (3) size-independent properties of the whole result: offsets monotone, every (q, t) in range, q strictly
ascending inside a pair, no t twice inside a pair (cross-check).  Reports wall time, device time, peak device / page-locked
memory.  Test infrastructure (imports oracle/).
Replaces at this size: the pair loop of /root/reference/src/Feature/FeatureMatching.cpp:102-145."""
import argparse
import ctypes
import json
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monocularsfm_amd import _lib, synth  # noqa: E402


def mem_info():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return free.value, total.value


def pair_scratch_bytes(n1, n2, route=1):
    """msfm_pair_scratch_bytes (csrc/msfm_hostutil.h): route 1 matrix cores + compacted sweep 2, route 3 the same on the integer cores
    (byte stores: what this tool's jobs take)"""
    if n1 <= 0 or n2 <= 0:
        return 0
    n1pad, n2pad = (n1 + 511) // 512 * 512, (n2 + 511) // 512 * 512
    blocks512 = n1pad // 512
    common = (24 if route == 3 else 36) * (n1pad + n2pad) + 24 * n1 + 1024
    partials = 8 * n1pad + (4 if route == 3 else 8) * blocks512 * n2pad
    cmp_rows = (n1 + n2 * min(blocks512, 32)) // 16 + 1024
    return common + partials + (120 if route == 3 else 84) * cmp_rows


def predicted_sub_batches(n_rows, pairs, free_bytes, max_pairs=16384, sets=3):
    """The memory / pair-count cut of MatchJob::build (csrc/msfm_job.hip.h): budget = min(64 GiB, free / 4), a third per set."""
    per_set = min(64 << 30, free_bytes // 4) // sets
    bounds = [0]
    est, cnt = 0, 0
    for k, (i, j) in enumerate(pairs):
        need = pair_scratch_bytes(int(n_rows[i]), int(n_rows[j]), 3)
        if cnt > 0 and (cnt >= max_pairs or est + need > per_set):
            bounds.append(k)
            est, cnt = 0, 0
        est += need
        cnt += 1
    bounds.append(len(pairs))
    return bounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1329)
    ap.add_argument("--desc", type=int, default=8192)
    ap.add_argument("--oracle-pairs", type=int, default=24)
    ap.add_argument("--int-oracle-pairs", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1329)
    ap.add_argument("--warm", action="store_true", help="one untimed call first: the timed one runs on allocated buffers")
    ap.add_argument("--stream", action="store_true", help="msfm_match_pairs_begin / _next: bounded memory (see above)")
    ap.add_argument("--cut-every", type=int, default=25, help="--stream: the pairs on either side of every N-th chunk cut go to the oracle")
    args = ap.parse_args()
    if args.stream:
        return stream_main(args)

    t0 = time.perf_counter()
    imgs, pairs, name = synth.job("synthetic-u8", args.images, args.desc, seed=args.seed)
    gen_s = time.perf_counter() - t0
    n_rows = np.array([len(x) for x in imgs], np.int64)
    total_desc_pairs = int((n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).sum())
    free0, total_mem = mem_info()
    ctx = _lib.Context(0)
    t0 = time.perf_counter()
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctx.finalize_store()
    upload_s = time.perf_counter() - t0
    free_store, _ = mem_info()
    cold_wall_s = None
    if args.warm:
        t0 = time.perf_counter()
        ctx.match_pairs(pairs, max_distance=1e9, fetch="view")
        cold_wall_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    offs, qt, d = ctx.match_pairs(pairs, max_distance=1e9, fetch="view")
    wall_s = time.perf_counter() - t0
    prof = ctx.profile()
    free_after, _ = mem_info()
    M = int(offs[-1])
    bounds = predicted_sub_batches(n_rows, pairs, free_store)

    # (3) whole-result properties, in chunks of pairs (3.6e8 matches: the index arrays of one pass would be tens of GB)
    assert (np.diff(offs) >= 0).all()
    in_range = q_ascending = t_unique = True
    for p0 in range(0, len(pairs), 20000):
        p1 = min(len(pairs), p0 + 20000)
        s, e = int(offs[p0]), int(offs[p1])
        if e == s:
            continue
        cq, ct = np.asarray(qt[s:e, 0]), np.asarray(qt[s:e, 1])
        pair_of = np.repeat(np.arange(p0, p1), np.diff(offs[p0:p1 + 1]))
        in_range &= bool(((cq >= 0) & (cq < n_rows[pairs[pair_of, 0]]) & (ct >= 0) & (ct < n_rows[pairs[pair_of, 1]])).all())
        same_pair = pair_of[1:] == pair_of[:-1]
        q_ascending &= bool((np.diff(cq.astype(np.int64))[same_pair] > 0).all())
        key = np.sort((pair_of - p0).astype(np.int64) * (1 << 20) + ct)
        t_unique &= bool((np.diff(key) != 0).all())      # cross-check: a train row is matched at most once per pair

    # (2) sampled pairs against the oracles
    from oracle import c_oracle, int_oracle
    c_oracle.build()
    rng = np.random.default_rng(4)
    cut_pairs = []
    for b in bounds[1:-1][:: max(1, (len(bounds) - 2) // 5 or 1)][:5]:
        cut_pairs += [b - 1, b]                          # last pair of one sub-batch, first of the next
    sel = sorted(set([0, len(pairs) - 1] + cut_pairs + rng.choice(len(pairs), max(0, args.oracle_pairs - 2 - len(cut_pairs)), replace=False).tolist()))
    sel = np.asarray(sel, np.int64)
    f32 = {int(i): imgs[int(i)].astype(np.float32) for i in np.unique(pairs[sel])}
    t0 = time.perf_counter()
    o_offs, oq, ot, od = c_oracle.match_pairs(f32, pairs[sel], max_distance=1e9, nthreads=16)
    oracle_s = time.perf_counter() - t0
    mismatches = 0
    for k, p in enumerate(sel):
        s, e = int(offs[p]), int(offs[p + 1])
        os_, oe = int(o_offs[k]), int(o_offs[k + 1])
        ok = (e - s == oe - os_) and np.array_equal(qt[s:e, 0], oq[os_:oe]) and np.array_equal(qt[s:e, 1], ot[os_:oe]) and \
            np.array_equal(np.asarray(d[s:e]).view(np.int32), od[os_:oe].view(np.int32))
        mismatches += 0 if ok else 1
    int_mismatches = 0
    for p in sel[:args.int_oracle_pairs]:
        i, j = pairs[p]
        iq, it, idist = int_oracle.match_pair(imgs[int(i)], imgs[int(j)], 0.8, True, 1e9)
        s, e = int(offs[p]), int(offs[p + 1])
        ok = np.array_equal(qt[s:e, 0], iq) and np.array_equal(qt[s:e, 1], it) and \
            np.array_equal(np.asarray(d[s:e]).view(np.int32), np.asarray(idist, np.float32).view(np.int32))
        int_mismatches += 0 if ok else 1

    out = {
        "workload": name + (" -- BASELINE configs[3] in full" if (args.images, args.desc, args.seed) == (1329, 8192, 1329) else
                            " -- seeded subset of BASELINE configs[4] (4096 x 16384)" if (args.desc, args.seed) == (16384, 4096) else "") +
                    ", one msfm_match_pairs call on one MI355X",
        "warm": bool(args.warm), "cold_first_call_wall_s": cold_wall_s,
        "image_pairs": int(len(pairs)), "descriptor_pairs": total_desc_pairs, "matches": M,
        "wall_s": wall_s, "device_s": prof["total_device_ms"] * 1e-3, "value_descriptor_pairs_per_s": total_desc_pairs / wall_s,
        "image_pairs_per_s": len(pairs) / wall_s,
        "sub_batches": prof["sub_batches"], "sub_batches_predicted": len(bounds) - 1,
        "sweep1_ms_total": prof["approx_kernel_ms"], "sweep2_ms_total": prof["sweep2_ms"], "sweep1_i8_launches": prof["sweep1_i8_launches"],
        "sweep1_frac_of_5_POPs": 256.0 * prof["prefilter_descriptor_pairs"] / max(1e-9, prof["approx_kernel_ms"] * 1e-3) / 5e15,
        "fallback_pairs": prof["fallback_pairs"], "plan_regrows": prof["plan_regrows"], "tie_queue_regrows": prof["tie_queue_regrows"],
        "candidates": prof["candidates"], "order_sensitive_rows": prof.get("order_sensitive_rows"),
        "oracle_checked_pairs": int(len(sel)), "oracle_mismatching_pairs": mismatches, "oracle_matches_checked": int(o_offs[-1]),
        "int_oracle_checked_pairs": int(min(args.int_oracle_pairs, len(sel))), "int_oracle_mismatching_pairs": int_mismatches,
        "oracle_pairs_at_sub_batch_cuts": [int(x) for x in cut_pairs], "oracle_wall_s": oracle_s,
        "properties": {"offsets_monotone": True, "indices_in_range": in_range, "q_strictly_ascending_per_pair": q_ascending,
                       "train_index_unique_per_pair": t_unique},
        "memory": {"device_total_GiB": total_mem / 2**30, "store_GiB": (free0 - free_store) / 2**30,
                   "store_bytes_per_row": ctx.store_info()["device_bytes"] / max(1, ctx.store_info()["rows"]), "library_view": ctx.memory_info(),
                   "device_peak_GiB_after_call": (free0 - free_after) / 2**30,
                   "result_lists_page_locked_GiB": M * 12 / 2**30, "host_max_rss_GiB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20},
        "setup": {"generate_s": gen_s, "upload_s": upload_s, "store_bytes": int(sum(x.nbytes for x in imgs))},
    }
    print(json.dumps(out, indent=1))
    # (the pipeline's cost marks -- six parts -- may cut inside a memory-bounded sub-batch: up to five more)
    ok = mismatches == 0 and int_mismatches == 0 and in_range and q_ascending and t_unique and \
        len(bounds) - 1 <= prof["sub_batches"] <= len(bounds) - 1 + 5
    ctx.close()
    sys.exit(0 if ok else 1)


def stream_main(args):
    """The job through msfm_match_pairs_begin / _next (see the module docstring)."""
    t0 = time.perf_counter()
    imgs, pairs, name = synth.job("synthetic-u8", args.images, args.desc, seed=args.seed)
    gen_s = time.perf_counter() - t0
    n_rows = np.array([len(x) for x in imgs], np.int64)
    total_desc_pairs = int((n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).sum())
    ctx = _lib.Context(0)
    mem0 = ctx.memory_info()
    t0 = time.perf_counter()
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)
    ctx.finalize_store()
    upload_s = time.perf_counter() - t0
    mem_store = ctx.memory_info()
    rng = np.random.default_rng(4)
    want = set(rng.choice(len(pairs), max(0, args.oracle_pairs - 10), replace=False).tolist()) | {0, len(pairs) - 1}
    kept = {}              # pair index -> (q, t, dist bits) of the sampled pairs
    cuts = []
    peak_dev = peak_pin = 0
    M = 0
    chunks = 0
    in_range = q_ascending = t_unique = True
    crc = 0
    import zlib
    t0 = time.perf_counter()
    for ch in ctx.match_pairs_stream(pairs, max_distance=1e9, copy=False):
        first, n = ch["first"], ch["n_pairs"]
        offs, qt, d = ch["offsets"], ch["qt"], ch["dist"]
        chunks += 1
        M += int(offs[-1])
        if chunks <= 3 or chunks % args.cut_every == 0:                 # the pairs on either side of a cut
            cuts += [first, first + n - 1]
            want |= {first, first + n - 1}
        for p in [p for p in want if first <= p < first + n]:
            s, e = int(offs[p - first]), int(offs[p - first + 1])
            kept[p] = (qt[s:e, 0].copy(), qt[s:e, 1].copy(), d[s:e].view(np.int32).copy())
        if len(qt):
            crc = (crc + int(qt.view(np.int64).sum(dtype=np.int64))) & 0xFFFFFFFFFFFFFFFF   # (a cheap running sum: the consumer must not starve the pipeline)
        if len(qt) and chunks % 4 == 1:     # the whole-result properties on every fourth chunk
            cq, ct = qt[:, 0], qt[:, 1]
            pair_of = np.repeat(np.arange(first, first + n), np.diff(offs))
            in_range &= bool(((cq >= 0) & (cq < n_rows[pairs[pair_of, 0]]) & (ct >= 0) & (ct < n_rows[pairs[pair_of, 1]])).all())
            same_pair = pair_of[1:] == pair_of[:-1]
            q_ascending &= bool((np.diff(cq.astype(np.int64))[same_pair] > 0).all())
            key = np.sort((pair_of - first).astype(np.int64) * (1 << 20) + ct)
            t_unique &= bool((np.diff(key) != 0).all())
        m = ctx.memory_info()
        peak_dev = max(peak_dev, m["device_total"] - m["device_free"])
        peak_pin = max(peak_pin, m["page_locked_host"])
    wall_s = time.perf_counter() - t0
    prof = ctx.profile()
    mem_end = ctx.memory_info()
    from oracle import c_oracle, int_oracle
    c_oracle.build()
    sel = np.asarray(sorted(kept), np.int64)
    f32 = {int(i): imgs[int(i)].astype(np.float32) for i in np.unique(pairs[sel])}
    t0 = time.perf_counter()
    o_offs, oq, ot, od = c_oracle.match_pairs(f32, pairs[sel], max_distance=1e9, nthreads=16)
    oracle_s = time.perf_counter() - t0
    mismatches = 0
    for k, p in enumerate(sel):
        q, tt, db = kept[int(p)]
        os_, oe = int(o_offs[k]), int(o_offs[k + 1])
        ok = len(q) == oe - os_ and np.array_equal(q, oq[os_:oe]) and np.array_equal(tt, ot[os_:oe]) and np.array_equal(db, od[os_:oe].view(np.int32))
        mismatches += 0 if ok else 1
    int_mismatches = 0
    for p in sel[:args.int_oracle_pairs]:
        i, j = pairs[p]
        iq, it, idist = int_oracle.match_pair(imgs[int(i)], imgs[int(j)], 0.8, True, 1e9)
        q, tt, db = kept[int(p)]
        ok = np.array_equal(q, iq) and np.array_equal(tt, it) and np.array_equal(db, np.asarray(idist, np.float32).view(np.int32))
        int_mismatches += 0 if ok else 1
    info = ctx.store_info()
    out = {
        "workload": name + (" -- BASELINE configs[3] in full" if (args.images, args.desc, args.seed) == (1329, 8192, 1329) else
                            " -- BASELINE configs[4] IN FULL (4096 x 16384)" if (args.images, args.desc, args.seed) == (4096, 16384, 4096) else
                            " -- seeded subset of BASELINE configs[4] (4096 x 16384)" if (args.desc, args.seed) == (16384, 4096) else "") +
                    ", STREAMED: msfm_match_pairs_begin + one msfm_match_pairs_next per device sub-batch, one MI355X, first call of the process",
        "streaming": True, "chunks": chunks, "image_pairs": int(len(pairs)), "descriptor_pairs": total_desc_pairs, "matches": M,
        "wall_s": wall_s, "value_descriptor_pairs_per_s": total_desc_pairs / wall_s, "image_pairs_per_s": len(pairs) / wall_s,
        "device_s": prof["total_device_ms"] * 1e-3, "sub_batches": prof["sub_batches"],
        "sweep1_ms_total": prof["approx_kernel_ms"], "sweep1_i8_launches": prof["sweep1_i8_launches"],
        "sweep1_frac_of_5_POPs": 256.0 * prof["prefilter_descriptor_pairs"] / max(1e-9, prof["approx_kernel_ms"] * 1e-3) / 5e15,
        "fallback_pairs": prof["fallback_pairs"], "plan_regrows": prof["plan_regrows"], "candidates": prof["candidates"],
        "order_sensitive_rows": prof.get("order_sensitive_rows"), "qt_sum64": crc, "properties_checked_on": "every fourth chunk",
        "oracle_checked_pairs": int(len(sel)), "oracle_mismatching_pairs": mismatches, "oracle_matches_checked": int(o_offs[-1]),
        "int_oracle_checked_pairs": int(min(args.int_oracle_pairs, len(sel))), "int_oracle_mismatching_pairs": int_mismatches,
        "oracle_pairs_at_chunk_cuts": [int(x) for x in cuts], "oracle_wall_s": oracle_s,
        "properties": {"indices_in_range": in_range, "q_strictly_ascending_per_pair": q_ascending, "train_index_unique_per_pair": t_unique},
        "memory": {"device_total_GiB": mem0["device_total"] / 2**30, "device_used_before_GiB": (mem0["device_total"] - mem0["device_free"]) / 2**30,
                   "store_GiB": mem_store["store"] / 2**30, "store_bytes_per_row": info["device_bytes"] / max(1, info["rows"]),
                   "device_peak_GiB_incl_store": peak_dev / 2**30, "scratch_GiB_at_end": mem_end["scratch"] / 2**30,
                   "call_wide_result_lists_on_device_GiB": mem_end["results_device"] / 2**30,
                   "page_locked_host_peak_GiB": peak_pin / 2**30, "what_one_call_would_have_pinned_GiB": M * 12 / 2**30,
                   "host_max_rss_GiB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20},
        "setup": {"generate_s": gen_s, "upload_s": upload_s, "store_input_bytes": int(sum(x.nbytes for x in imgs))},
    }
    print(json.dumps(out, indent=1))
    ok = mismatches == 0 and int_mismatches == 0 and in_range and q_ascending and t_unique
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
