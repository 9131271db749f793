"""`python bench.py --gpus N` without a launcher in the command starts its own ranks (CPU check: no GPU here, so every rank
must stop with the no-fallback message -- which proves the ranks were started -- and the launcher returns their failure)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_torchrun_starts_two_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU box: covered by tests/test_bench_contract.py::test_bench_gpus_2_launches_itself")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--images", "4", "--u8-images", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "needs torch.distributed.run" not in r.stderr          # (round 3's refusal)
    assert r.stderr.count("no GPU visible (there is no CPU fallback)") >= 1, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
