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
