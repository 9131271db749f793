#!/usr/bin/env python3
"""bench.py -- ComputeMatches hot path on MI355X: descriptor-pairs/s (and image-pairs/s).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or plain
   `python bench.py --gpus N`: without WORLD_SIZE in the environment the script launches that command itself, one rank per
   GPU, passes rank 0's JSON line through and returns the job's exit code)

Workload (BASELINE.json configs[1]): South-Building-shaped job -- 128 images, ~5k 128-D float32 RootSIFT-like
descriptors each (seeded synthetic), brute-force all pairs (8128 image pairs, pre-emptive filter
off), reference defaults ratio 0.8 / cross-check / 0.7.  One "step" = the whole job: every pair through distance +
kNN-2 in both directions + ratio + cross-check + distance cut, with the match lists in host memory on the writer
rank at the end (N > 1: per-pair counts all-reduced, the lists sent over RCCL from HBM to the writer).
Descriptors are resident in HBM before the timed region.

STRONG scaling: the SAME job at every N (the N = 1 line is the BENCH line); the pair list is cut into N contiguous
cost-balanced ranges, the store is replicated.  `strong_u8` in the same JSON line is the strong-scaling measurement
north_star's >= 6x target is stated on: BASELINE configs[3] IN FULL (1329 images x 8192 u8 descriptors, 882 456 pairs,
5.92e13 descriptor pairs; ~9 s per step on one GPU), one warm-up step + `--u8-steps` (default 1) timed steps at every N --
strong_u8.value at N = 8 over the value at N = 1 IS that measurement.  `--u8-images M` (< 1329) runs a seeded M-image subset
instead (tests), 0 skips the job.  `--workload synthetic-u8 --images 1329 --desc 8192` runs it as the main workload.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (sweep 1 of the MFMA prefilter), timed with
HIP events on the library's own stream inside the timed region; `cpu_baseline` is the CPU oracle ("port": a
restatement of the OpenCV BFMatcher path, not OpenCV itself) on a bounded sample of the same pairs, on this box's
host cores, parallel over image pairs.
"""
import argparse
import glob
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
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak
PEAK_I8_MFMA_TOPS = 5000.0      # dense int8 MFMA (2 x K of fp16)


def pmc_traffic(i8):
    """HBM bytes per sweep-1 launch from the committed rocprofv3 PMC passes of this same command (separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of `bench.py --u8-images 0`; newest profiles/r*_pmc_traffic_approx.json): KB -> bytes,
    FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64 B, MI355X_MICROARCH.md).  The dominant kernel is
    sweep_i8_kernel<1> when sweep 1 ran on the integer cores (byte stores, byte twins of float stores), else
    sweep_kernel<1>.  None if absent."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_approx.json")), reverse=True):
        try:
            d = json.load(open(path))
            # steps of the profiled command (`--steps 3 --warmup 2` = 5 in every committed pass; "_meta" since round 4)
            pmc_steps = float(d.get("_meta", {}).get("steps", 5))
            d = {n: k for n, k in d.items() if not n.startswith("_")}
            per = {n: 2.0 * k["FETCH_SIZE"]["per_launch_KB_mean"] * 1024 + k["WRITE_SIZE"]["per_launch_KB_mean"] * 1024
                   for n, k in d.items()}
            want = ("sweep_i8_kernel<1>",) if i8 else ("approx_kernel<1>", "sweep1", "sweep_kernel<1>")
            name = [n for n in per if n.startswith(want)][0]
            # the profiled command may cut a step into another number of launches than this run: bytes per STEP are comparable
            launches = float(d[name]["FETCH_SIZE"].get("launches", pmc_steps))
            return {"bytes_per_step": per[name] * launches / pmc_steps, "pmc_launches_per_step": launches / pmc_steps,
                    "bytes_per_pmc_launch": per[name], "kernel": name, "source": os.path.relpath(path, ROOT),
                    "per_kernel_bytes_per_pmc_launch": per}
        except (OSError, KeyError, ValueError, IndexError):
            continue
    return None


def result_checksum(result):
    import zlib
    offs, qt = result[0], result[1]
    c = zlib.crc32(np.ascontiguousarray(offs, dtype=np.int64).tobytes())
    if qt is not None:
        c = zlib.crc32(np.ascontiguousarray(qt, dtype=np.int32).tobytes(), c)
    return int(c)


def opencv_found():
    try:
        import cv2
        return cv2.__version__
    except Exception:
        return False


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut by the cgroup CPU quota (a container on a 256-thread
    host is often given far fewer; os.cpu_count() reports the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(round(quota))))
    return n, (os.cpu_count() or 1), quota


def cpu_baseline(imgs, pairs, budget_s=20.0, max_pairs=4096, seed=0):
    """CPU oracle on a seeded sample of the same pairs: one persistent pool over image pairs on all host threads
    (orc_match_pairs_mt), plus the single-thread figure (how the reference itself runs: one pair after the other)."""
    from oracle import c_oracle as co
    co.build()
    threads, host_threads, quota = usable_cores()
    rng = np.random.default_rng(seed)
    order = rng.permutation(len(pairs))
    f32 = {}

    def need(sel):
        for i in np.unique(pairs[sel]):
            if int(i) not in f32:
                f32[int(i)] = np.ascontiguousarray(imgs[int(i)], dtype=np.float32)

    def work(sel):
        return int(sum(len(imgs[i]) * len(imgs[j]) for i, j in pairs[sel]))

    # single thread (how the reference itself runs: one pair after the other)
    one = order[:1]
    need(one)
    t0 = time.perf_counter()
    co.match_pairs(f32, pairs[one], nthreads=1)
    dt1 = time.perf_counter() - t0
    single = work(one) / dt1
    # all threads: the seeded order is worked off until the budget is spent (no new pair is started after it)
    sample = order[:min(max_pairs, len(pairs))]
    need(sample)
    # (the budget is shared: 65 % for the SSE order -- the `value` -- and 35 % for the fastest fused order this host runs on real intrinsics)
    t0 = time.perf_counter()
    offs, _, _, _, n = co.match_pairs(f32, pairs[sample], nthreads=threads, budget_s=0.65 * budget_s, return_done=True)
    dt = time.perf_counter() - t0
    w = work(sample[:n])
    # The baseline bounded from the FAST side: a real OpenCV on this host would dispatch its AVX2 / AVX-512
    # normL2Sqr_, not the SSE loop.  The oracle's fused orders run on real intrinsics here (oracle/msfm_oracle.c, bit-identical to
    # their plain-C statements): the same pool, the same sample order, the fastest of them.
    fast = None
    try:
        import ctypes as C
        L = co.lib()
        L.orc_simd_level.restype = C.c_int
        level = int(L.orc_simd_level())
        cands = [(co.ORDER_AVX512_FMA, "avx512-fma") if level & 2 else None, (co.ORDER_AVX2_FMA, "avx2-fma") if level & 1 else None]
        for cand in [c for c in cands if c]:
            t1 = time.perf_counter()
            _, _, _, _, nf = co.match_pairs(f32, pairs[sample], nthreads=threads, budget_s=0.35 * budget_s / max(1, len([c for c in cands if c])),
                                            order=cand[0], return_done=True)
            dtf = time.perf_counter() - t1
            rate = work(sample[:nf]) / dtf
            if fast is None or rate > fast["value"]:
                fast = {"value": rate, "order": cand[1], "image_pairs": int(nf), "wall_s": dtf}
    except Exception as e:  # noqa: BLE001  (an oracle build without the intrinsics: the SSE figure stands alone)
        fast = {"error": "%s: %s" % (type(e).__name__, e)}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": w / dt, "unit": "descriptor-pairs/s", "cores": threads, "kind": "port",
        "sample": "%d of %d image pairs (seeded), both kNN-2 sweeps + ratio + cross-check + distance cut per pair, %.1f s "
                  "wall; restated CPU BFMatcher (SSE order), persistent pthread pool over image pairs, not OpenCV" % (n, len(pairs), dt),
        "image_pairs_per_s": n / dt, "matches_in_sample": int(offs[-1]),
        "fast_order_value": None if not fast or "value" not in fast else fast["value"], "fast_order": fast,
        "single_thread_value": single, "parallel_efficiency": (w / dt) / (single * threads), "cpu_model": model,
        "host_hardware_threads": host_threads, "cgroup_cpu_quota": quota,
        "opencv_found": opencv_found(),
    }


def end_to_end_ratios(out, cpu_rate, cpu_single_rate):
    """The CPU side of the end_to_end block, once this run's cpu_baseline is known."""
    if cpu_rate and "descriptor_pairs" in out:
        total, wall = out["descriptor_pairs"], out["wall_s"]
        out["cpu_port_matching_s_estimate"] = total / cpu_rate
        out["cpu_port_single_thread_matching_s_estimate"] = total / cpu_single_rate if cpu_single_rate else None
        out["ratio"] = out["cpu_port_matching_s_estimate"] / wall
        out["ratio_vs_single_thread"] = (total / cpu_single_rate) / wall if cpu_single_rate else None
    return out


def end_to_end(cpu_rate, cpu_single_rate, n_images=128, n_desc=5000):
    """"wall-clock ... includes DB I/O for the ComputeMatches end-to-end figure; report both".  north_star states its
    >= 10x target on the ComputeMatches wall clock.  Writes the South-Building-shaped SQLite database (outside every timed region), then
    runs the drop-in executable `monocularsfm_amd/host/ComputeMatches <yaml>` cold -- a fresh process: HIP start-up, context,
    every allocation, the bulk load, pre-emptive filter, matching, geometric verification on the device, the rows written -- the way a
    user runs it (reference: sfm/ComputeMatches.cpp:59-65 prints the same wall clock); three such processes, the median reported.  The CPU side is an ESTIMATE of the matching
    alone (no DB I/O, no RANSAC) from this run's cpu_baseline rate: a lower bound of what the reference's CLI would take here."""
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    from monocularsfm_amd import synth
    exe = os.path.join(ROOT, "monocularsfm_amd", "host", "ComputeMatches")
    if not os.path.exists(exe):
        return {"error": "monocularsfm_amd/host/ComputeMatches is not built"}
    tmp = tempfile.mkdtemp(prefix="msfm_bench_e2e_")
    try:
        db_path = os.path.join(tmp, "south-building-synth.db")
        t0 = time.perf_counter()
        descs, _ = synth.south_building_database(db_path, n_images, n_desc, seed=1234)
        build_s = time.perf_counter() - t0
        cfg = os.path.join(tmp, "cfg.yaml")
        open(cfg, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\n' % db_path)
        env = dict(os.environ, MSFM_CLI_TIMING="1")
        # THREE runs, each in a fresh process on an untouched copy of the database (a run leaves its rows behind), each cold in every
        # sense the process controls: HIP start-up, context, every allocation, the bulk load.  wall_s is their MEDIAN, all three are
        # listed: the runtime's start-up alone is 85 ms or 160-240 ms from one process to the next on the same box
        # (profiles/r05_hip_init_settle.txt), and the first GPU process on a box that has just been handed over also pays the device's
        # wake-up (round 5, one box: 0.60 s against 0.41 s).  Every run starts on a device that has been left alone for a second: a process
        # that initialises the runtime within ~0.1 s of another GPU process's exit waits for the driver to finish tearing that one down
        # (hipGetDeviceCount 170-240 ms instead of 52 ms, with this executable and with a one-line HIP program alike).
        settle_s = 1.0
        runs = []
        for k in range(3):
            db_k = db_path if k == 0 else db_path + ".%d" % k
            cfg_k = os.path.join(tmp, "cfg%d.yaml" % k)
            if k > 0:
                shutil.copyfile(db_path + ".pristine", db_k)
            else:
                shutil.copyfile(db_path, db_path + ".pristine")
            open(cfg_k, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\n' % db_k)
            time.sleep(settle_s)
            t0 = time.perf_counter()
            r = subprocess.run([exe, cfg_k], capture_output=True, text=True, env=env, timeout=600)
            wall_k = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "ComputeMatches exited with %d: %s" % (r.returncode, r.stderr[-300:])}
            runs.append((wall_k, r, db_k))
        walls = [w for w, _, _ in runs]
        wall, r, db_used = sorted(runs, key=lambda x: x[0])[1]
        # a fourth process with the geometric verification OFF: what the matcher hands to FeatureUtils::FilterMatches
        # (src/Feature/FeatureUtils.cpp:176-206) -- the synthetic keypoints observe shared scene points (synth.scene_keypoints), so the
        # verification keeps the true matches as it does on overlapping photographs, and the write phase is a representative one
        noverify = None
        try:
            db_n = db_path + ".noverify"
            shutil.copyfile(db_path + ".pristine", db_n)
            cfg_n = os.path.join(tmp, "cfg_noverify.yaml")
            open(cfg_n, "w").write('%%YAML:1.0\ndatabase_path : "%s"\nSIFTmatch.match_type : 1\n' % db_n)
            time.sleep(settle_s)
            t0 = time.perf_counter()
            rn = subprocess.run([exe, cfg_n], capture_output=True, text=True, env=dict(env, MSFM_GEOMETRIC_VERIFICATION="0"), timeout=600)
            wall_n = time.perf_counter() - t0
            if rn.returncode == 0:
                con = sqlite3.connect(db_n)
                rows_n, matches_n = con.execute("SELECT COUNT(*), SUM(rows) FROM matches").fetchone()
                con.close()
                noverify = {"wall_s": wall_n, "rows_written": int(rows_n), "matches_written": int(matches_n or 0)}
        except Exception as e:  # noqa: BLE001
            noverify = {"error": "%s: %s" % (type(e).__name__, e)}
        phases = {}
        for name, val in re.findall(r"([a-zA-Z+\- ]+?) ([0-9.]+) s(?: \||$)", ([l for l in r.stderr.splitlines() if "exist-check" in l] or [""])[-1].replace("[msfm timing] ", "")):
            phases[name.strip()] = float(val)
        con = sqlite3.connect(db_used)
        rows, matches = con.execute("SELECT COUNT(*), SUM(rows) FROM matches").fetchone()
        con.close()
        n_rows = np.array([len(d) for d in descs], np.int64)
        pairs = n_images * (n_images - 1) // 2
        total = int((n_rows.sum() ** 2 - (n_rows ** 2).sum()) // 2)
        out = {"command": "monocularsfm_amd/host/ComputeMatches <yaml> (brute-force mode, pre-emptive filter and geometric verification on: the reference's defaults)",
               "wall_s": wall, "cold": True, "walls_s": walls, "wall_s_is": "the median of three fresh processes (walls_s, in the order they ran); phases_s: that run's",
               "phases_s": phases, "phases_sum_s": sum(phases.values()), "settle_s_before_each_process": settle_s,
               "db_bytes": os.path.getsize(db_path), "db_build_s_untimed": build_s, "images": n_images, "pairs": pairs,
               "rows_written": int(rows), "matches_written": int(matches or 0), "descriptor_pairs": total,
               "verification_off": noverify,
               "matches_kept_by_verification": (int(matches or 0) / noverify["matches_written"]) if noverify and noverify.get("matches_written") else None,
               "matches_written_per_pair": int(matches or 0) / max(1, int(rows)),
               "file_cache": "warm (the database was written just before the run)",
               "last_stdout_line": (r.stdout.strip().splitlines() or [""])[-1]}
        return end_to_end_ratios(out, cpu_rate, cpu_single_rate)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="south-building", choices=["south-building", "synthetic-u8"])
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
                    help="brute-force exact-order kernel for every pair (same results, ~20x slower)")
    ap.add_argument("--f16-only", action="store_true", help="byte stores on the fp16 matrix cores too (default: integer matrix cores)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--u8-images", type=int, default=1329,
                    help="images of the strong-scaling job on the 1329 x 8192 u8 config (default: the config in full; fewer = "
                         "a seeded subset); 0 = skip")
    ap.add_argument("--u8-steps", type=int, default=None, help="timed steps of that job (default 1 for the full config, else 2)")
    ap.add_argument("--sustained-steps", type=int, default=200,
                    help="extra untimed-for-`value` run of this many steps after the K timed ones -> sustained_ms_per_step "
                         "(the part's clock is set by a power budget: a 1 s burst and a 10 s run differ); 0 = skip")
    ap.add_argument("--super-batch-pairs", type=int, default=0,
                    help="bounded memory: the step's lists are produced and consumed chunk by chunk -- N = 1: the streaming form of the C ABI "
                         "(msfm_match_pairs_begin / _next, one device sub-batch per chunk; the value only switches it on); N > 1: "
                         "ShardedMatcher.match_to_writer_batches with super-batches of this many pairs.  0 = one call, lists resident")
    ap.add_argument("--no-e2e", action="store_true", help="skip the cold end-to-end run of the ComputeMatches executable (end_to_end)")
    ap.add_argument("--no-solo", action="store_true",
                    help="skip the 4 extra steps with the pipeline off that measure the sweeps alone (roofline.solo); for kernel traces")
    args = ap.parse_args()
    if args.u8_steps is None:
        args.u8_steps = 1 if args.u8_images >= 1329 else 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run, rank 0's JSON line
        # (the only thing the ranks print on stdout) passes through, the job's exit code is ours
        import subprocess
        # --standalone: torchrun's own rendezvous picks a free port itself (binding a socket, closing it and handing the
        # number on left a window for another process to take the port); 127.0.0.1: the container's hostname may not resolve
        env = dict(os.environ)
        env.pop("MASTER_PORT", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "2")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        # (the ranks print one thing on stdout -- rank 0's JSON line -- but libraries under them may not keep to that: gloo
        # announces its connections there; anything that is not the line goes to stderr)
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for line in proc.stdout:
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        raise SystemExit(proc.wait())

    # The cold end-to-end run of the drop-in executable comes FIRST, before this process touches the GPU: a user runs ComputeMatches
    # alone.  (Run behind the benchmark -- this process idle, but holding its HIP queues and ~60 GB of scratch -- the same command took
    # 0.79 s instead of 0.41 s on one box, 0.34 s of it in the matching call: two processes' queues on one GPU.)
    e2e = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_e2e and args.workload == "south-building" and not args.force_collectives:
        try:
            e2e = end_to_end(None, None, n_images=args.images or 128)
        except Exception as e:  # noqa: BLE001  (a failure here must not take the headline line with it)
            e2e = {"error": "%s: %s" % (type(e).__name__, e)}

    import torch
    import torch.distributed as dist
    from monocularsfm_amd import _lib, synth
    from monocularsfm_amd.sharding import ShardedMatcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world   # (under a launcher the process group is what counts)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no GPU visible (there is no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live
    multi = world > 1 or args.force_collectives
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ranks_seen = None
    if multi:   # every rank of the process group answers: an all_reduce of ones over the exchange backend (RCCL when --backend nccl)
        one = torch.ones(1, dtype=torch.int32, device=coll_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.cpu()[0])
    ctx = _lib.Context(local_rank, order=args.order)
    if args.no_prefilter:
        ctx.set_prefilter(False)
    elif args.f16_only:
        ctx.set_prefilter(2)

    def run_job(imgs, pairs, steps, warmup, collect=None, sustained_steps=0, solo_pass=True, **match_kw):
        """Upload, W untimed + K timed steps; -> (seconds of the K steps: max over ranks, result of the last step,
        upload seconds, per-rank [compute_ms, exchange_ms] means)."""
        n_rows = np.array([len(x) for x in imgs], np.int64)
        ctx.clear_images()
        t_up = time.perf_counter()
        for i, im in enumerate(imgs):
            ctx.upload_image(i, im)   # resident in HBM before the timed region
        ctx.finalize_store()          # (uploads only copy; the images are built -- classification, layout kernels -- here, inside the timer)
        upload_s = time.perf_counter() - t_up   # host buffers -> HBM (PCIe) + the on-device layout / fp16 / norm passes
        sm = ShardedMatcher(ctx=ctx, device=coll_dev, force_collectives=args.force_collectives, **match_kw)

        stream_stats = {"chunks": 0, "max_chunk_matches": 0, "page_locked_peak": 0, "device_peak": 0}

        def step_streamed():
            """The lists never exist as a whole: every chunk is consumed (a running sum over its rows: what a database writer or an
            RCCL send would do with it) and dropped."""
            counts = np.zeros(len(pairs), np.int64)
            acc = [0]
            t = time.perf_counter()
            if not multi:
                for ch in ctx.match_pairs_stream(pairs, copy=False, **match_kw):
                    counts[ch["first"]:ch["first"] + ch["n_pairs"]] = np.diff(ch["offsets"])
                    if len(ch["qt"]):
                        acc[0] = (acc[0] + int(ch["qt"].view(np.int64).sum(dtype=np.int64))) & 0xFFFFFFFFFFFFFFFF
                    stream_stats["chunks"] += 1
                    stream_stats["max_chunk_matches"] = max(stream_stats["max_chunk_matches"], len(ch["qt"]))
                sm.last = {"compute_ms": (time.perf_counter() - t) * 1e3, "exchange_ms": 0.0}
            else:
                def sink(b0, offs, qt, d):
                    if len(qt):
                        acc[0] = (acc[0] + int(np.ascontiguousarray(qt).view(np.int64).sum(dtype=np.int64))) & 0xFFFFFFFFFFFFFFFF
                    stream_stats["chunks"] += 1
                    stream_stats["max_chunk_matches"] = max(stream_stats["max_chunk_matches"], len(qt))
                counts = sm.match_to_writer_batches(pairs, n_rows, batch_pairs=args.super_batch_pairs, dst=0, sink=sink)
            m = ctx.memory_info()
            stream_stats["page_locked_peak"] = max(stream_stats["page_locked_peak"], m["page_locked_host"])
            stream_stats["device_peak"] = max(stream_stats["device_peak"], m["device_total"] - m["device_free"])
            stream_stats["qt_sum64"] = acc[0]
            offs = np.zeros(len(pairs) + 1, np.int64)
            np.cumsum(counts, out=offs[1:])
            return offs, None, None

        def step():
            if args.super_batch_pairs > 0:
                return step_streamed()
            if not multi:
                # one rank: the step ends with the lists in the library's page-locked host buffers ("view": no second copy)
                t = time.perf_counter()
                offs, qt, _ = ctx.match_pairs(pairs, fetch="view", **match_kw)
                sm.last = {"compute_ms": (time.perf_counter() - t) * 1e3, "exchange_ms": 0.0}
                return offs, qt, None
            return sm.match_to_writer(pairs, n_rows, dst=0, with_dist=False)

        result = None
        for _ in range(warmup):
            result = step()
        barrier()
        phases = np.zeros(2)
        t0 = time.perf_counter()
        for _ in range(steps):
            result = step()
            phases += (sm.last["compute_ms"], sm.last["exchange_ms"])
            if collect is not None:
                collect(ctx.profile())
        barrier()
        dt = time.perf_counter() - t0
        sustained = None
        if sustained_steps > 0:   # after the timed region: the same step, long enough for the power / thermal state to settle
            t1 = time.perf_counter()
            for _ in range(sustained_steps):
                result = step()
            barrier()
            sustained = (time.perf_counter() - t1) / sustained_steps
        per_rank = [(phases / steps).tolist()]
        if world > 1:
            t = torch.tensor([dt] + (phases / steps).tolist(), dtype=torch.float64, device=coll_dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            allt = torch.stack(allt).cpu().numpy()
            dt = float(allt[:, 0].max())
            per_rank = allt[:, 1:].tolist()
            if sustained is not None:
                t = torch.tensor([sustained], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sustained = float(t.cpu()[0])
        run_job.sustained = sustained
        # the last step's exchange as this rank saw it: the library's device-to-device copy into the (torch-allocated) send tensor and its
        # rate -- the cross-runtime pointer of monocularsfm_amd/_lib.py -- and, for streamed steps, how much of the exchange the matching
        # thread actually waited for (sharding.match_to_writer_batches pipelines it under the next super-batch)
        last = dict(sm.last)
        run_job.exchange_detail = None if not multi else {
            "fetch_device_ms": last.get("fetch_device_ms"), "fetch_device_bytes": last.get("fetch_device_bytes"),
            "fetch_device_GBps": (last["fetch_device_bytes"] / 1e9 / (last["fetch_device_ms"] * 1e-3)) if last.get("fetch_device_ms") else None,
            "exchange_ms": last.get("exchange_ms"), "exchange_wait_ms": last.get("exchange_wait_ms"), "pipelined": last.get("pipelined"),
            "local_matches": last.get("local_matches"), "super_batches": last.get("super_batches")}
        run_job.stream_stats = dict(stream_stats) if args.super_batch_pairs > 0 else None
        # after everything timed: the sweeps ALONE (one sub-batch per step, nothing in flight beside them) -- in the timed
        # region the other sub-batches' bandwidth-bound tails run beside a sweep and stretch its event span
        run_job.solo = None
        if collect is not None and solo_pass and not args.no_prefilter and not args.no_solo and os.environ.get("MSFM_PIPELINE") is None:
            ctx.set_pipeline(1)
            solo = {"approx_kernel_ms": 0.0, "approx_kernel_launches": 0, "sweep2_ms": 0.0, "prefilter_descriptor_pairs": 0, "wall_ms": 0.0}
            try:
                step()
                for _ in range(3):
                    t2 = time.perf_counter()
                    step()
                    solo["wall_ms"] += (time.perf_counter() - t2) * 1e3 / 3
                    p = ctx.profile()
                    for k in ("approx_kernel_ms", "approx_kernel_launches", "sweep2_ms", "prefilter_descriptor_pairs"):
                        solo[k] += p[k]
            finally:
                ctx.set_pipeline(0)
            barrier()
            run_job.solo = solo
        return dt, result, upload_s, per_rank, n_rows

    # ---- main workload -----------------------------------------------------------------------------------------
    imgs, pairs, wl_name = synth.job(args.workload, args.images, args.desc, seed=args.seed)
    main_kw = {"max_distance": 1e9} if args.workload == "synthetic-u8" else {}
    acc = {k: 0 for k in ("dist_kernel_ms", "dist_kernel_launches", "approx_kernel_ms", "approx_kernel_launches",
                          "prefilter_descriptor_pairs", "exact_descriptor_pairs", "candidates", "fallback_pairs", "sweep2_ms",
                          "sweep2_launches", "sweep2_descriptor_pairs", "compacted_pairs", "total_device_ms", "sub_batches",
                          "sweep1b_ms", "sweep1b_launches", "sweep1b_descriptor_pairs", "sweep1_q8_launches")}
    last_prof = {}

    def collect(p):
        for k in acc:
            acc[k] += p[k]
        last_prof.update(p)

    dt, result, upload_s, per_rank, n_rows = run_job(imgs, pairs, args.steps, args.warmup, collect, args.sustained_steps, **main_kw)
    sustained = run_job.sustained
    solo = run_job.solo
    stream_main = run_job.stream_stats
    main_exchange_detail = run_job.exchange_detail
    total_desc_pairs = int((n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).sum())
    offs = result[0]
    n_matches = int(offs[-1])
    value = total_desc_pairs * args.steps / dt
    u8_store = imgs[0].dtype == np.uint8
    out = {
        "metric": "descriptor-pairs/sec (and image-pairs/sec); match-index bit-parity vs CPU",
        "value": value, "unit": "descriptor-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "path": "brute-force exact fp32" if args.no_prefilter else "MFMA prefilter + exact fp32 re-check (bit-identical results)",
        "image_pairs_per_s": len(pairs) * args.steps / dt,
        # not `value`: one-time store upload (PCIe + layout kernels) added to one step (which already
        # includes the result copy-out to host memory)
        "pcie_inclusive": {"upload_ms": upload_s * 1e3, "store_bytes": int(sum(x.nbytes for x in imgs)),
                           "value_incl_upload": total_desc_pairs / (upload_s + dt / args.steps)},
        "config": {"workload": wl_name, "image_pairs": int(len(pairs)), "descriptor_pairs_per_step": total_desc_pairs,
                   "matches_per_step": n_matches, "accum_order": "opencv-sse4x4-nofma" if args.order == 0 else "opencv-avx2-fma",
                   "ratio": 0.8, "cross_check": True, "max_distance": main_kw.get("max_distance", 0.7), "preemptive_filter": False,
                   "parallelism": "the SAME job at every N: image pairs cut into %d contiguous cost-balanced range(s), store "
                                  "replicated; exchange = all_reduce of per-pair counts + RCCL send of the (q, t) lists from HBM "
                                  "to the writer rank" % world},
        # this rank's view per step (ms): matcher call (sweeps + epilogue + copy into the send buffer) vs exchange
        "per_rank_ms": [{"compute": c, "exchange": e} for c, e in per_rank],
        "exchange_detail_rank0": main_exchange_detail,
        "device_ms_per_step_rank0": acc["total_device_ms"] / args.steps,
        # the same step repeated --sustained-steps times right after the timed region (power / thermal steady state)
        "sustained_ms_per_step": None if sustained is None else sustained * 1e3,
        "sustained_steps": args.sustained_steps,
        "sustained_value": None if sustained is None else total_desc_pairs / sustained,
        # CRC-32 of the writer rank's (offsets, (q, t) rows) of the last step: equal across N and across exchange paths
        "exchange_checksum": result_checksum(result) if rank == 0 else None,
        # rows / columns of this rank's pairs whose decisions are NOT certified order-invariant (msfm_fetch_order_certificate):
        # 0 => the index lists are the same under any conforming fp32 order of OpenCV's normL2Sqr_
        "order_sensitive_rows": int(last_prof.get("order_sensitive_rows", -1)),
        "sub_batches_per_step": acc["sub_batches"] // max(1, args.steps),
        # the step is cut into sub-batches (two equal ones for this job) launched on separate streams: the bandwidth-bound tail of
        # one runs beside the sweeps of the next (msfm_set_pipeline / MSFM_PIPELINE; 1 = one launch per sweep, no overlap)
        "pipeline_env": os.environ.get("MSFM_PIPELINE"),
        # ranks that answered an all_reduce over the exchange backend before the job started (None at N = 1 without --force-collectives)
        "rccl_ranks_seen" if args.backend == "nccl" else "gloo_ranks_seen": ranks_seen,
        "exchange_backend": args.backend if multi else None,
        # --super-batch-pairs: the lists of a step were produced and consumed chunk by chunk (bounded memory)
        "streamed": None if not stream_main else dict(stream_main, super_batch_pairs=args.super_batch_pairs,
                                                     page_locked_peak_GiB=stream_main["page_locked_peak"] / 2**30,
                                                     device_peak_GiB=stream_main["device_peak"] / 2**30),
    }
    pf_ms, pf_launches = acc["approx_kernel_ms"], acc["approx_kernel_launches"]
    if pf_launches > 0:
        # dominant kernel of the default path: sweep 1 of the prefilter (every descriptor pair of the rank's range once).
        # Sweep 2 only revisits the rows / columns the ratio and distance tests left alive.
        i8 = bool(last_prof.get("sweep1_i8_launches", 0))
        peak = PEAK_I8_MFMA_TOPS if i8 else PEAK_F16_MFMA_TFLOPS
        pf_pairs_work = acc["prefilter_descriptor_pairs"]
        avg_ms = pf_ms / pf_launches
        flops = 256.0 * pf_pairs_work   # GEMM form: 128 x (mul, add) per descriptor pair
        achieved = flops / (pf_ms * 1e-3) / 1e12
        tr = pmc_traffic(i8)
        algo_bytes_step = last_prof.get("dist_algo_bytes", 0)
        rows_work = float((n_rows[pairs[:, 0]] + n_rows[pairs[:, 1]]).sum()) * args.steps / max(world, 1)
        # ONE launch unit for `traffic`, `algorithmic_bytes_per_launch` and `achieved`: this run's average sweep-1 launch
        # (the step is cut into `launches_per_step` of them).  The PMC passes are separate runs of the same command -- counters
        # cannot be collected inside the timed process -- so their bytes per STEP are divided by this run's launches per step.
        launches_per_step = pf_launches / float(args.steps)
        traffic = tr["bytes_per_step"] / launches_per_step if tr else None
        algo_launch = algo_bytes_step / launches_per_step
        # what the sweep itself reads when nothing is re-used: both images' operand rows once per image pair -- 176-byte byte-twin
        # rows on the integer cores (128 operand bytes + digits + constants), 272-byte fp16 rows otherwise
        row_bytes = 176.0 if i8 else 272.0
        algo_rows_launch = float((n_rows[pairs[:, 0]] + n_rows[pairs[:, 1]]).sum()) / max(world, 1) * row_bytes / launches_per_step
        out["roofline"] = {
            "kernel": "sweep 1 of the MFMA prefilter (%s%s); sweep 2 and the exact fp32 re-check only touch survivors"
                      % ("v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_f16",
                         ", on the byte twins of the float store: route Q" if acc["sweep1_q8_launches"] else ""),
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
            "frac": achieved / peak,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch of THIS run (PMC bytes per step / launches per step)",
            "traffic_detail": tr,
            # not measured by THIS run: rocprofv3 --pmc passes cannot run inside the timed process
            "traffic_source": ("committed PMC pass of this command: " + tr["source"]) if tr else None,
            "avg_launch_ms": avg_ms, "launches": pf_launches, "launches_per_step": launches_per_step, "flops_per_desc_pair": 256.0,
            "descriptor_pairs_per_launch": pf_pairs_work / pf_launches,
            # (n1 + n2) x 128 x 4 B + the kNN lists per image pair, no cross-pair reuse -- per launch, like `traffic`
            "algorithmic_bytes_per_launch": algo_launch,
            "traffic_over_algorithmic": (traffic / algo_launch) if (traffic is not None and algo_launch > 0) else None,
            "algorithmic_bytes_per_launch_operand_rows": algo_rows_launch,
            "traffic_over_algorithmic_operand_rows": (traffic / algo_rows_launch) if (traffic is not None and algo_rows_launch > 0) else None,
            "traffic_per_step": tr["bytes_per_step"] if tr else None, "algorithmic_bytes_per_step": algo_bytes_step,
            # (event spans: with sub-batches in flight a sweep-2 launch waits for another stream's sweep 1 --
            # that wait is inside its span; profiles/rNN_route_q_kernel_stats_pipeline1.txt has the unpipelined kernel times)
            "sweep2": {"ms_per_step": acc["sweep2_ms"] / args.steps, "launches": acc["sweep2_launches"],
                       "compacted_image_pairs": acc["compacted_pairs"] // max(1, args.steps),
                       "work_fraction_of_sweep1": acc["sweep2_descriptor_pairs"] / max(1, pf_pairs_work)},
            # route Q (float store, byte twins): sweep 1 above runs on the twins; sweep 1' (coarse twins only) is the fp16 sweep of the rows it left alive
            "route_q": {"twin_sweep_launches": acc["sweep1_q8_launches"], "sweep1b_ms_per_step": acc["sweep1b_ms"] / args.steps,
                        "sweep1b_work_fraction_of_sweep1": acc["sweep1b_descriptor_pairs"] / max(1, pf_pairs_work)},
            # the same kernel with nothing beside it: 3 further steps with the pipeline off (one launch per sweep), after the timed region
            "solo": None if not solo or not solo["approx_kernel_launches"] else {
                "avg_launch_ms": solo["approx_kernel_ms"] / solo["approx_kernel_launches"], "launches": solo["approx_kernel_launches"],
                "achieved": 256.0 * solo["prefilter_descriptor_pairs"] / (solo["approx_kernel_ms"] * 1e-3) / 1e12,
                "frac": 256.0 * solo["prefilter_descriptor_pairs"] / (solo["approx_kernel_ms"] * 1e-3) / 1e12 / peak,
                "sweep2_ms_per_step": solo["sweep2_ms"] / 3, "ms_per_step_unpipelined": solo["wall_ms"]},
            "sweep1_ms_per_step": pf_ms / args.steps,
            "step_over_sweep1": (dt / args.steps * 1e3) / max(1e-9, pf_ms / args.steps),
            "candidates_per_row": acc["candidates"] / max(1.0, rows_work),
            "fallback_pairs": acc["fallback_pairs"],
            "hbm": {"algorithmic_bytes_per_step": algo_bytes_step,
                    "achieved_GBps": (algo_bytes_step * args.steps / (pf_ms * 1e-3)) / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                    "frac": (algo_bytes_step * args.steps / (pf_ms * 1e-3)) / 1e9 / PEAK_HBM_GBPS},
        }
    if acc["dist_kernel_launches"] > 0:
        avg_ms = acc["dist_kernel_ms"] / acc["dist_kernel_launches"]
        flops_per_launch = FLOPS_PER_DESC_PAIR * acc["exact_descriptor_pairs"] / acc["dist_kernel_launches"]
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        exact = {
            "kernel": "dist_top2_kernel (brute-force exact-order path)", "bound": "valu", "achieved": achieved,
            "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": None,
            "avg_launch_ms": avg_ms, "launches": acc["dist_kernel_launches"], "flops_per_desc_pair": FLOPS_PER_DESC_PAIR,
            "note": "384 unfusable sub/mul/add per descriptor pair: 78.6 TFLOP/s (half the FMA peak) is the "
                    "attainable ceiling; frac_of_nofma_ceiling reports against that",
            "frac_of_nofma_ceiling": achieved / (PEAK_FP32_VALU_TFLOPS / 2),
        }
        if "roofline" in out:
            out["roofline_exact_path"] = exact
        else:
            out["roofline"] = exact

    # ---- the strong-scaling job of north_star's >= 6x target: BASELINE configs[3] (1329 x 8192 u8), in full by default --------
    def u8_job():
        full = args.u8_images >= 1329
        n_img = min(args.u8_images, 1329)
        t_gen = time.perf_counter()
        if world > 1:
            # N ranks: the set is generated ONCE (rank 0; 13 s for the full config, and the ranks share a 16-CPU cgroup) and reaches the
            # others through a file under $TMPDIR, read back memory-mapped
            import tempfile
            cache = os.path.join(tempfile.gettempdir(), "msfm_u8_%dx8192_seed1329.npy" % n_img)
            if rank == 0 and not os.path.exists(cache):
                u_imgs, u_pairs, u_name = synth.job("synthetic-u8", n_img, 8192, seed=1329)
                tmp_path = cache + ".%d.tmp.npy" % os.getpid()
                np.save(tmp_path, np.stack(u_imgs))
                os.replace(tmp_path, cache)
            dist.barrier()
            stack = np.load(cache, mmap_mode="r")
            u_imgs = [stack[i] for i in range(n_img)]
            u_pairs = synth.all_pairs(n_img)
            u_name = "synthetic u8 descriptors: %d images x 8192 desc, brute-force all pairs" % n_img
        else:
            u_imgs, u_pairs, u_name = synth.job("synthetic-u8", n_img, 8192, seed=1329)
        gen_s = time.perf_counter() - t_gen
        u_acc = {"approx_kernel_ms": 0.0, "prefilter_descriptor_pairs": 0, "sweep1_i8_launches": 0, "sub_batches": 0,
                 "approx_kernel_launches": 0, "order_sensitive_rows": 0, "candidates": 0}

        def u_collect(p):
            for k in u_acc:
                u_acc[k] += p.get(k, 0)

        # u8 distances are hundreds, not fractions of a unit-norm descriptor: no distance cut (reference default 0.7 is for RootSIFT).
        # One warm-up step (buffers sized, plan hints learnt), then the timed ones; the sweeps-alone pass only on small subsets
        # (three more steps of the full job would be half a minute)
        u_dt, u_res, u_upload_s, u_per_rank, u_rows = run_job(u_imgs, u_pairs, args.u8_steps, 1, u_collect, solo_pass=n_img <= 256,
                                                              max_distance=1e9)
        u_solo = run_job.solo
        u_total = int((u_rows[u_pairs[:, 0]] * u_rows[u_pairs[:, 1]]).sum())
        i8 = u_acc["sweep1_i8_launches"] > 0
        u_peak = PEAK_I8_MFMA_TOPS if i8 else PEAK_F16_MFMA_TFLOPS
        u_ach = 256.0 * u_acc["prefilter_descriptor_pairs"] / max(1e-9, u_acc["approx_kernel_ms"] * 1e-3) / 1e12
        u_val = u_total * args.u8_steps / u_dt
        return {
            "workload": u_name + (" -- BASELINE configs[3] IN FULL" if full else
                                  " (seeded subset of BASELINE configs[3]: %d of 1329 images, full per-image size)" % n_img),
            "full_config": full,
            "value": u_val, "unit": "descriptor-pairs/s", "ms_per_step": u_dt / args.u8_steps * 1e3,
            "seconds_per_step": u_dt / args.u8_steps, "image_pairs_per_s": len(u_pairs) * args.u8_steps / u_dt,
            "steps": args.u8_steps, "warmup": 1, "image_pairs": int(len(u_pairs)), "descriptor_pairs_per_step": u_total,
            "matches_per_step": int(u_res[0][-1]), "n_gpus": world, "scaling": "strong",
            "per_rank_ms": [{"compute": c, "exchange": e} for c, e in u_per_rank],
            "exchange_detail_rank0": run_job.exchange_detail,
            "sub_batches_per_step_rank0": u_acc["sub_batches"] // max(1, args.u8_steps),
            "order_sensitive_rows_rank0": int(u_acc["order_sensitive_rows"]) // max(1, args.u8_steps),
            "sweep1": {"instruction": "v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_f16", "achieved": u_ach,
                       "frac": u_ach / u_peak, "ms_per_step_rank0": u_acc["approx_kernel_ms"] / args.u8_steps,
                       "step_over_sweep1_rank0": (u_dt / args.u8_steps * 1e3) / max(1e-9, u_acc["approx_kernel_ms"] / args.u8_steps),
                       # the kernel with nothing beside it (pipeline off, after the timed steps; subsets only)
                       "solo_frac": None if not u_solo or not u_solo["approx_kernel_ms"] else
                       256.0 * u_solo["prefilter_descriptor_pairs"] / (u_solo["approx_kernel_ms"] * 1e-3) / 1e12 / u_peak},
            "setup": {"generate_s": gen_s, "upload_s": u_upload_s, "store_bytes": int(sum(x.nbytes for x in u_imgs))},
            "full_config_seconds_at_this_rate": 882456 * 8192.0 * 8192.0 / u_val,
        }

    if args.u8_images > 1 and args.workload == "south-building" and not args.no_prefilter:
        if world == 1:
            try:   # (a failure of the second job must not take the headline line with it)
                out["strong_u8"] = u8_job()
            except Exception as e:  # noqa: BLE001
                out["strong_u8"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            out["strong_u8"] = u8_job()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(imgs, pairs, budget_s=args.cpu_budget)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        out["gpu_over_cpu_fast_order"] = value / max(out["cpu_baseline"]["value"], out["cpu_baseline"].get("fast_order_value") or 0.0)
    if e2e is not None:
        cb = out.get("cpu_baseline", {})
        # (against the FASTER of the port's orders: the ratio is a lower bound)
        out["end_to_end"] = end_to_end_ratios(e2e, max(cb.get("value") or 0.0, cb.get("fast_order_value") or 0.0) or None, cb.get("single_thread_value"))
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
