"""CPU tier: bench.py's command line (the driver's contract: --gpus N --steps K --warmup W) and its refusal to run
without a GPU -- the product path has no CPU fallback, so the benchmark must fail loudly instead of measuring one."""
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_help_lists_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


def test_bench_fails_loudly_without_a_gpu():
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                  # no JSON line: nothing was measured
    assert "no CPU fallback" in r.stderr or "HIP" in r.stderr
