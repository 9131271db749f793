"""The BASELINE byte configs at (or near) full size on one MI355X, oracle-sampled -- opt-in: `python -m pytest tests -m slow` on the GPU
box (minutes; NOT part of `-m gpu`; skipped without a GPU).  The runs behind profiles/r05_config4_full.json and
profiles/r05_config5_1024_stream.json, as tests a driver can execute ("the only oracle-checked runs of config 4 IN FULL
... are builder-run").  Each is tools/config4_full.py in a process of its own: exit status 0 = no oracle mismatch among the pairs
sampled at the sub-batch cuts and at random, whole-result properties hold.  Loop these shard:
/root/reference/src/Feature/FeatureMatching.cpp:102-145."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.slow


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:   # noqa: BLE001
        return False


def _run(args, timeout):
    if not _gpu():
        pytest.skip("needs an MI355X")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "config4_full.py")] + args, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(r.stdout[r.stdout.index("{"):])


def test_config4_in_full_one_call():
    d = _run(["--int-oracle-pairs", "1"], 1200)
    assert d["image_pairs"] == 882456 and d["oracle_mismatching_pairs"] == 0 and d["int_oracle_mismatching_pairs"] == 0
    assert d["order_sensitive_rows"] == 0 and d["fallback_pairs"] == 0 and all(d["properties"].values())
    assert d["memory"]["store_bytes_per_row"] <= 0.4 * 1024


def test_config5_at_1024_images_streamed():
    d = _run(["--images", "1024", "--desc", "16384", "--seed", "4096", "--stream", "--int-oracle-pairs", "1"], 1800)
    assert d["image_pairs"] == 523776 and d["oracle_mismatching_pairs"] == 0 and d["int_oracle_mismatching_pairs"] == 0
    assert d["memory"]["device_peak_GiB_incl_store"] <= 60.0 and d["memory"]["page_locked_host_peak_GiB"] <= 2.0
    assert d["memory"]["call_wide_result_lists_on_device_GiB"] == 0.0


def test_config5_at_512_images_one_call():
    d = _run(["--images", "512", "--desc", "16384", "--seed", "4096", "--int-oracle-pairs", "1"], 1200)
    assert d["image_pairs"] == 130816 and d["oracle_mismatching_pairs"] == 0 and all(d["properties"].values())


def test_config5_in_full_streamed():
    """BASELINE configs[4] IN FULL (4096 images x 16384 byte descriptors: 8 386 560 pairs, 2.25e15 descriptor pairs, 6.9e9 matches)
    through msfm_match_pairs_begin / _next on ONE GPU: ~5.5 minutes of matching + ~1.5 minutes of generation + the oracle on the sampled
    pairs (profiles/r06_config5_full_stream.json: 319.8 s, 7.04e12 descriptor-pairs/s, device peak 62.1 GiB, 0.15 GiB page-locked)."""
    d = _run(["--images", "4096", "--desc", "16384", "--seed", "4096", "--stream", "--oracle-pairs", "24", "--int-oracle-pairs", "1",
              "--cut-every", "100"], 3000)
    assert d["image_pairs"] == 8386560 and d["oracle_checked_pairs"] >= 24
    assert d["oracle_mismatching_pairs"] == 0 and d["int_oracle_mismatching_pairs"] == 0 and all(d["properties"].values())
    assert d["order_sensitive_rows"] == 0 and d["fallback_pairs"] == 0
    assert d["memory"]["device_peak_GiB_incl_store"] <= 80.0 and d["memory"]["page_locked_host_peak_GiB"] <= 2.0


def test_executable_on_a_config4_shaped_database():
    """The drop-in EXECUTABLE at the scale the strong-scaling target is stated on: tools/cli_e2e_bench.py --config4 on the byte side table
    (1329 images x 8192 descriptors, verification off), rows in the reference's order and in pair_id order.  Every pair gets its row, both
    orders write the same number of matches, and the run is a pipeline: wall <= 1.15 x max(device threads, emission) + what comes
    before and after the matching phase (profiles/r06_cli_config4.txt: 15.7 s emission-bound / 9.7 s device-bound)."""
    if not _gpu():
        pytest.skip("needs an MI355X")
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="msfm_slow_"), "cli_config4.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_e2e_bench.py"), "--config4", "--tables", "u8", "--modes", "off",
                        "--json", out], capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = json.load(open(out))
    runs = {x["emit_order"]: x for x in d["runs"]}
    assert set(runs) == {"reference", "pair_id"}
    for x in runs.values():
        assert x["rows_written"] == x["image_pairs"] == 882456 and x["stdout_lines"] >= 3 * 882456
        ph = x["phases_s"]
        around = sum(ph[k] for k in ("exist-check", "read descriptors + upload", "pre-emptive filter", "open database + device", "close"))
        assert x["wall_s"] <= 1.15 * max(ph["device match + fetch"], ph["stdout + WriteMatches"]) + around + 1.0, x
    assert runs["reference"]["matches_written"] == runs["pair_id"]["matches_written"] > 3e8
    assert runs["pair_id"]["phases_s"]["stdout + WriteMatches"] < 0.5 * runs["reference"]["phases_s"]["stdout + WriteMatches"]
