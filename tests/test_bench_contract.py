"""bench.py's one-line JSON contract (driver + judge read it): required keys, types and internal consistency, on a
reduced workload so the test stays short."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--images", "24", "--cpu-budget", "2", "--u8-images", "6"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["scaling"] == "strong" and isinstance(d["config"].get("workload"), str)
    assert len(d["per_rank_ms"]) == 1 and d["per_rank_ms"][0]["compute"] > 0
    assert d["strong_u8"]["value"] > 0 and d["strong_u8"]["image_pairs"] == 15
    assert "opencv_found" in d["cpu_baseline"] and d["cpu_baseline"]["parallel_efficiency"] > 0
    assert "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = descriptor pairs of the job / step time
    assert abs(d["value"] - d["config"]["descriptor_pairs_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma", "valu") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # `traffic`, `algorithmic_bytes_per_launch` and `achieved` refer to ONE launch unit: this run's average sweep-1 launch
    lps = rf["launches_per_step"]
    assert abs(lps - rf["launches"] / d["steps"]) < 1e-9
    assert abs(rf["algorithmic_bytes_per_launch"] * lps - rf["algorithmic_bytes_per_step"]) < 1e-6 * rf["algorithmic_bytes_per_step"]
    assert abs(rf["descriptor_pairs_per_launch"] * lps - d["config"]["descriptor_pairs_per_step"]) < 1e-6 * d["config"]["descriptor_pairs_per_step"]
    assert rf["algorithmic_bytes_per_launch_operand_rows"] < rf["algorithmic_bytes_per_launch"]
    if rf["traffic"] is not None:
        assert "per launch of THIS run" in rf["traffic_unit"]
        assert abs(rf["traffic"] * lps - rf["traffic_per_step"]) < 1e-6 * rf["traffic_per_step"]
        assert abs(rf["traffic_over_algorithmic"] - rf["traffic"] / rf["algorithmic_bytes_per_launch"]) < 1e-9
        assert abs(rf["traffic_over_algorithmic"] - rf["traffic_per_step"] / rf["algorithmic_bytes_per_step"]) < 1e-6
    su = d["strong_u8"]
    assert su["full_config"] is False and su["n_gpus"] == 1 and su["scaling"] == "strong" and su["sweep1"]["frac"] > 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    # SURVEY 8(d) "report both": the cold end-to-end run of the drop-in executable next to the kernel figure
    e2e = d["end_to_end"]
    assert "error" not in e2e, e2e
    for k in ("wall_s", "phases_s", "db_bytes", "pairs", "rows_written", "cold", "cpu_port_matching_s_estimate", "ratio"):
        assert k in e2e, k
    assert e2e["cold"] is True and e2e["pairs"] == 24 * 23 // 2 and 0 < e2e["rows_written"] <= e2e["pairs"]
    assert e2e["wall_s"] > 0 and 0 < e2e["phases_sum_s"] <= e2e["wall_s"] and "device match + fetch" in e2e["phases_s"]
    assert len(e2e["walls_s"]) == 3 and sorted(e2e["walls_s"])[1] == e2e["wall_s"]     # three fresh processes, the median is the figure
    assert abs(e2e["ratio"] - e2e["cpu_port_matching_s_estimate"] / e2e["wall_s"]) < 1e-9
    assert d["pcie_inclusive"]["upload_ms"] > 0


def test_bench_two_ranks_sharing_the_gpu():
    """The multi-rank flow of bench.py (partition, per-rank match, count all_reduce, send of the lists to the writer)
    with two ranks on this box's one GPU over gloo: RCCL itself needs a GPU per rank (driver's SCALE run)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--images", "20", "--u8-images", "5", "--backend", "gloo", "--share-gpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and len(d["per_rank_ms"]) == 2
    assert d["config"]["image_pairs"] == 190 and d["config"]["matches_per_step"] > 0
    assert all(x["compute"] > 0 for x in d["per_rank_ms"]) and "cpu_baseline" not in d
    # the same job on one rank gives the same number of matches
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--images", "20",
                         "--u8-images", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["config"]["matches_per_step"] == d["config"]["matches_per_step"]


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` WITHOUT torchrun in the command (the form of the driver's BENCH command with N changed): the
    script becomes the launcher, rank 0's JSON line is the only line on stdout, the exit code is the job's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--images", "20", "--u8-images", "5", "--backend", "gloo", "--share-gpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and len(d["per_rank_ms"]) == 2
    assert d["config"]["image_pairs"] == 190 and d["config"]["matches_per_step"] > 0
    su = d["strong_u8"]
    assert su["n_gpus"] == 2 and su["image_pairs"] == 10 and len(su["per_rank_ms"]) == 2 and su["value"] > 0
    # a failing job's exit code comes back through the launcher (more ranks than this box has GPUs, RCCL, no --share-gpu)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--images", "8", "--u8-images", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.device_count() < 2:
        assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_streamed_steps_give_the_same_job():
    """--super-batch-pairs: the step's lists are produced and consumed chunk by chunk (N = 1: msfm_match_pairs_begin / _next; N = 2:
    super-batches through the exchange) -- same number of matches as the one-call step, page-locked memory bounded by a chunk."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "synthetic-u8", "--images", "14", "--desc", "3000", "--steps", "2",
            "--warmup", "1", "--u8-images", "0", "--no-cpu-baseline", "--no-e2e", "--sustained-steps", "0", "--no-solo"]
    lines = {}
    for name, extra in (("one_call", []), ("streamed", ["--super-batch-pairs", "20"]),
                        ("two_ranks", ["--super-batch-pairs", "20", "--gpus", "2", "--backend", "gloo", "--share-gpu"])):
        # (N = 1: a chunk is a device sub-batch -- cut at 20 pairs here; N = 2: a chunk is a super-batch of the exchange)
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, MSFM_MAX_PAIRS_PER_BATCH="20"))
        assert r.returncode == 0, (name, r.stderr[-2000:])
        lines[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    m = lines["one_call"]["config"]["matches_per_step"]
    assert m > 100 and lines["one_call"]["streamed"] is None
    for name in ("streamed", "two_ranks"):
        d = lines[name]
        assert d["config"]["matches_per_step"] == m and d["streamed"]["chunks"] >= 2 * 5, d["streamed"]
        assert 0 < d["streamed"]["max_chunk_matches"] < m
    assert lines["streamed"]["streamed"]["qt_sum64"] == lines["two_ranks"]["streamed"]["qt_sum64"]
    assert lines["two_ranks"]["n_gpus"] == 2 and lines["two_ranks"]["gloo_ranks_seen"] == 2

