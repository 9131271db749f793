#!/usr/bin/env python3
"""bench.py -- ComputeMatches hot path on MI355X: descriptor-pairs/s (and image-pairs/s).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): South-Building-shaped job -- 128 images, ~5k 128-D float32
RootSIFT-like descriptors each (seeded synthetic), brute-force all pairs
(8128 image pairs, pre-emptive filter off), reference defaults ratio 0.8 / cross-check / 0.7.
One "step" = the whole job: every pair through distance + kNN-2 both directions + ratio +
cross-check + distance cut, match lists back on the host (and, for N > 1, all-gathered over RCCL).
Descriptors are resident in HBM before the timed region.  The same total job is split over the
ranks at N > 1 ("strong" scaling).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (dist_top2_kernel), timed
with HIP events on the library's own stream inside the timed region; `cpu_baseline` is the CPU
oracle ("port": a restatement of the OpenCV BFMatcher path, not OpenCV itself) on a bounded
sample of the same pairs, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_DESC_PAIR = 384.0   # 128 x (sub, mul, add), not fused (direct form)
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (FMA = 2 flop) = f32 MFMA rate
PEAK_HBM_GBPS = 8000.0
MEASURED_MFMA_TFLOPS = 4 * 256 * (2 * 32768) / 32.1e-9 / 1e12   # see roofline.measured_mfma_issue_ceiling
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak


def pmc_traffic():
    """HBM bytes per approx_kernel launch from the committed rocprofv3 PMC passes of this same command
    (separate --pmc FETCH_SIZE / WRITE_SIZE runs, profiles/r01_pmc_traffic_approx.json): KB -> bytes, and
    FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 B, MI355X_MICROARCH.md).  None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic_approx.json")
    try:
        d = json.load(open(path))
        per = {n: 2.0 * k["FETCH_SIZE"]["per_launch_KB_mean"] * 1024 + k["WRITE_SIZE"]["per_launch_KB_mean"] * 1024
               for n, k in d.items()}
        return {"bytes_per_launch": per["approx_kernel<1>"], "source": "profiles/r01_pmc_traffic_approx.json",
                "per_kernel_bytes": per}
    except (OSError, KeyError, ValueError):
        return None


def build_workload(args, world=1):
    from monocularsfm_amd import synth
    rng = np.random.default_rng(args.seed)
    if args.workload == "south-building":
        # weak scaling: the image set grows with sqrt(world) so that every GPU keeps ~8128 image pairs
        # (128 images at N=1 = BASELINE.json configs[1]; 181 / 256 / 362 images at N = 2 / 4 / 8)
        n_images = args.images or int(round(128 * np.sqrt(world)))
        counts = rng.integers(4600, 5401, n_images) if args.desc is None else np.full(n_images, args.desc)
        imgs = synth.rootsift_images(n_images, counts.tolist(), seed=args.seed, n_proto=20000, sigma=0.05)
        name = "south-building-shaped synthetic: %d images x ~%d f32 RootSIFT-like desc, brute-force all pairs" % (
            n_images, int(np.mean(counts)))
    elif args.workload == "synthetic-u8":
        n_images = args.images or 64
        nd = args.desc or 8192
        imgs = synth.u8_images(n_images, nd, seed=args.seed, as_float=True)
        name = "synthetic u8-valued: %d images x %d desc (f32-holding-integers), brute-force all pairs" % (n_images, nd)
    else:
        raise SystemExit("unknown workload " + args.workload)
    pairs = np.array([(i, j) for i in range(n_images) for j in range(i)], np.int32)
    return imgs, pairs, name


def cpu_baseline(imgs, pairs, budget_s=20.0, max_pairs=256, seed=0):
    """CPU oracle on a seeded sample of the same pairs, all host hardware threads."""
    from oracle import c_oracle as co
    co.build()
    threads = os.cpu_count() or 1
    rng = np.random.default_rng(seed)
    order = rng.permutation(len(pairs))
    # calibrate on one pair, then size the sample to the budget
    i, j = pairs[order[0]]
    t0 = time.perf_counter()
    co.match_pair(imgs[i], imgs[j], nthreads=threads)
    t1 = time.perf_counter() - t0
    n = int(max(4, min(max_pairs, budget_s / max(t1, 1e-4))))
    sample = order[:n]
    work = 0
    t0 = time.perf_counter()
    for p in sample:
        i, j = pairs[p]
        co.match_pair(imgs[i], imgs[j], nthreads=threads)
        work += len(imgs[i]) * len(imgs[j])
    dt = time.perf_counter() - t0
    # single-thread figure on a few pairs (the reference's own code is single-threaded; OpenCV's
    # batchDistance may fan out over its thread pool)
    k = min(2, n)
    t0 = time.perf_counter()
    w1 = 0
    for p in sample[:k]:
        i, j = pairs[p]
        co.match_pair(imgs[i], imgs[j], nthreads=1)
        w1 += len(imgs[i]) * len(imgs[j])
    dt1 = time.perf_counter() - t0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": work / dt, "unit": "descriptor-pairs/s", "cores": threads, "kind": "port",
        "sample": "%d of %d image pairs (seeded), both kNN-2 sweeps + ratio + cross-check + distance cut per pair, "
                  "%.1f s wall; restated CPU BFMatcher (SSE order, pthreads over query rows), not OpenCV" % (n, len(pairs), dt),
        "image_pairs_per_s": n / dt, "single_thread_value": w1 / dt1, "cpu_model": model,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="south-building")
    ap.add_argument("--images", type=int, default=None)
    ap.add_argument("--desc", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--order", type=int, default=0, help="0: OpenCV SSE order (default), 1: AVX2+FMA order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (+ --share-gpu) runs the multi-rank flow on a box with fewer GPUs than ranks")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses GPU 0 (flow check only: the value is meaningless)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="run the RCCL exchange step even with one rank (sanity check of the multi-GPU path on a 1-GPU box)")
    ap.add_argument("--no-prefilter", action="store_true",
                    help="brute-force exact-order kernel for every pair (same results, ~6x slower)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from monocularsfm_amd import _lib
    from monocularsfm_amd.sharding import ShardedMatcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no GPU visible (there is no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    imgs, pairs, wl_name = build_workload(args, world)
    n_rows = np.array([len(x) for x in imgs], np.int64)
    total_desc_pairs = int((n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).sum())

    ctx = _lib.Context(local_rank, order=args.order)
    if args.no_prefilter:
        ctx.set_prefilter(False)
    t_up = time.perf_counter()
    for i, im in enumerate(imgs):
        ctx.upload_image(i, im)   # resident in HBM before the timed region
    upload_s = time.perf_counter() - t_up   # host buffers -> HBM (PCIe) + the on-device layout / fp16 / norm passes
    # the step ends with the match lists in host memory (the library's page-locked result buffers; "view" = no second copy)
    sm = ShardedMatcher(ctx=ctx, device=coll_dev, force_collectives=args.force_collectives, fetch="view")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kern_ms, kern_launches, local_pairs_work = 0.0, 0, 0
    pf_ms, pf_launches, pf_pairs_work, exact_pairs_work, cand, rows_work, fallback = 0.0, 0, 0, 0, 0, 0, 0
    s2_ms, s2_launches, s2_pairs_work, compacted = 0.0, 0, 0, 0
    algo_bytes_step = 0
    result = None
    for _ in range(args.warmup):
        result = sm.match_to_writer(pairs, n_rows, dst=0, with_dist=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = sm.match_to_writer(pairs, n_rows, dst=0, with_dist=False)
        p = ctx.profile()
        kern_ms += p["dist_kernel_ms"]
        kern_launches += p["dist_kernel_launches"]
        local_pairs_work += p["descriptor_pairs"]
        algo_bytes_step = p["dist_algo_bytes"]
        pf_ms += p["approx_kernel_ms"]
        pf_launches += p["approx_kernel_launches"]
        pf_pairs_work += p["prefilter_descriptor_pairs"]
        exact_pairs_work += p["exact_descriptor_pairs"]
        cand += p["candidates"]
        fallback += p["fallback_pairs"]
        s2_ms += p["sweep2_ms"]
        s2_launches += p["sweep2_launches"]
        s2_pairs_work += p["sweep2_descriptor_pairs"]
        compacted += p["compacted_pairs"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    offs, qt, dd = result
    n_matches = int(offs[-1])
    rows_work = float((n_rows[pairs[:, 0]] + n_rows[pairs[:, 1]]).sum()) * args.steps / max(world, 1)

    value = total_desc_pairs * args.steps / dt
    out = {
        "metric": "descriptor-pairs/sec (and image-pairs/sec); match-index bit-parity vs CPU",
        "value": value, "unit": "descriptor-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "path": "brute-force exact fp32" if args.no_prefilter else "fp16 MFMA prefilter + exact fp32 re-check (bit-identical results)",
        "image_pairs_per_s": len(pairs) * args.steps / dt,
        # not `value`: one-time store upload (PCIe + layout kernels) added to one step (which already
        # includes the result copy-out to host memory)
        "pcie_inclusive": {"upload_ms": upload_s * 1e3, "store_bytes": int(sum(x.nbytes for x in imgs)),
                           "value_incl_upload": total_desc_pairs / (upload_s + dt / args.steps)},
        "config": {"workload": wl_name, "image_pairs": int(len(pairs)), "descriptor_pairs_per_step": total_desc_pairs,
                   "matches_per_step": n_matches, "accum_order": "opencv-sse4x4-nofma" if args.order == 0 else "opencv-avx2-fma",
                   "ratio": 0.8, "cross_check": True, "max_distance": 0.7, "preemptive_filter": False,
                   "parallelism": "image pairs sharded over %d GPU(s) (contiguous cost-balanced ranges, store replicated); "
                                  "exchange = all_reduce of per-pair counts + gather of the (q, t) lists to the writer rank" % world},
    }
    if pf_launches > 0:
        # dominant kernel of the default path: approx_kernel<1> (MFMA fp16 32x32x16), sweep 1: every descriptor
        # pair of the batch once.  Sweep 2 only revisits the rows / columns the ratio and distance tests left alive.
        avg_ms = pf_ms / pf_launches
        flops = 256.0 * pf_pairs_work   # GEMM form: 128 x (mul, add) per descriptor pair
        achieved = flops / (pf_ms * 1e-3) / 1e12
        tr = pmc_traffic()
        out["roofline"] = {
            "kernel": "approx_kernel<1> (MFMA prefilter sweep 1; sweep 2 and the exact fp32 re-check only touch survivors)",
            "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F16_MFMA_TFLOPS,
            # a loop of nothing but independent v_mfma_f32_32x32x16_f16 sustains 2 MFMA / 32.1 ns per SIMD on this part
            # (profiles/r01_ubench_mfma_valu.txt): 2.09 PFLOP/s, not the 2.5 of the data sheet
            "measured_mfma_issue_ceiling": MEASURED_MFMA_TFLOPS, "frac_of_measured_ceiling": achieved / MEASURED_MFMA_TFLOPS,
            "traffic": tr["bytes_per_launch"] if tr else None, "traffic_unit": "HBM bytes/launch (PMC)", "traffic_detail": tr,
            "avg_launch_ms": avg_ms, "launches": pf_launches, "flops_per_desc_pair": 256.0,
            "descriptor_pairs_per_launch": pf_pairs_work / pf_launches,
            "algorithmic_bytes_per_launch": algo_bytes_step * args.steps / pf_launches,
            "sweep2": {"ms_per_step": s2_ms / args.steps, "launches": s2_launches, "compacted_image_pairs": compacted // max(1, args.steps),
                       "work_fraction_of_sweep1": s2_pairs_work / max(1, pf_pairs_work)},
            "sweep1_ms_per_step": pf_ms / args.steps,
            "candidates_per_row": cand / max(1.0, rows_work),
            "fallback_pairs": fallback,
            "hbm": {"algorithmic_bytes_per_step": algo_bytes_step,
                    "achieved_GBps": (algo_bytes_step * args.steps / (pf_ms * 1e-3)) / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                    "frac": (algo_bytes_step * args.steps / (pf_ms * 1e-3)) / 1e9 / PEAK_HBM_GBPS},
        }
    if kern_launches > 0:
        avg_ms = kern_ms / kern_launches
        flops_per_launch = FLOPS_PER_DESC_PAIR * exact_pairs_work / kern_launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        exact = {
            "kernel": "dist_top2_kernel (brute-force exact-order path)", "bound": "valu", "achieved": achieved,
            "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": None,
            "avg_launch_ms": avg_ms, "launches": kern_launches, "flops_per_desc_pair": FLOPS_PER_DESC_PAIR,
            "note": "384 unfusable sub/mul/add per descriptor pair: 78.6 TFLOP/s (half the FMA peak) is the "
                    "attainable ceiling; frac_of_nofma_ceiling reports against that",
            "frac_of_nofma_ceiling": achieved / (PEAK_FP32_VALU_TFLOPS / 2),
        }
        if "roofline" in out:
            out["roofline_exact_path"] = exact
        else:
            out["roofline"] = exact
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(imgs, pairs, budget_s=args.cpu_budget)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
