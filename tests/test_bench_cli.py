"""bench.py's launcher contract without a GPU: it must fail loudly (exit code 2, nothing on stdout) instead of
falling back to a CPU path or to fewer GPUs than requested.  (The GPU-side cases live in
tests/test_gpu_sharded_rccl.py.)"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only contract")
@pytest.mark.parametrize("gpus", ["1", "2", "8"])
def test_bench_needs_a_gpu(gpus):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 2 and not out.stdout.strip()
    assert "MI355X" in out.stderr
