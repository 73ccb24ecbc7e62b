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


EXPECTED_OTHER = {"c2", "c4", "c5_per_gpu", "c5_full_1024_on_one_gpu", "add_c3_f32", "add_c3_bf16", "add_c4_f32",
                  "rnntloss_c3", "rnntloss_c5", "rnntloss_c2"}


def test_other_workloads_are_the_baseline_configs_and_the_additive_joint():
    """The default one-GPU line carries every BASELINE configuration and the additive joint:
    the key set is part of the contract with whoever reads BENCH_rNN.json."""
    sys.path.insert(0, ROOT)
    import bench
    assert set(bench.OTHER_WORKLOADS) | set(bench.ADD_WORKLOADS) | set(bench.MODULE_WORKLOADS) == EXPECTED_OTHER
    assert bench.WORKLOADS["c5_full"]["N"] == 1024 and bench.WORKLOADS["c5_full"]["dtype"] == "bf16"
    assert {bench.WORKLOADS[v]["dtype"] for v in bench.OTHER_WORKLOADS.values()} == {"fp32", "bf16"}


@pytest.mark.gpu
def test_default_line_carries_every_workload_with_a_passing_check():
    """`python bench.py --gpus 1` (few steps; the CPU baseline and the PMC passes are the headline's and are covered by the
    driver's own run): c3 stays value / roofline, `other_workloads` has the seven entries, each with ms_per_step, its
    stage times, a roofline fraction and a check against the oracle that PASSED."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-traffic-pass"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["config"]["workload"].startswith("c3") and rec["check"]["passed"] and rec["roofline"]["frac"] > 0.5
    assert rec["check"]["max_err_over_quantum"] <= 1.0 and rec["check"]["max_rel_grad_err"] <= 1e-3
    other = rec["other_workloads"]
    assert set(other) == EXPECTED_OTHER
    for key, e in other.items():
        assert e["ms_per_step"] > 0 and e["step_ms"]["p10"] <= e["step_ms"]["median"] <= e["step_ms"]["p90"], key
        assert (e.get("stage_ms") or key.startswith("rnntloss_")) and e["path_frac"] > 0, key
        assert e["check"]["passed"], (key, e["check"])
        # the per-element keys: error / (one rounding of the stored value + fp32 arithmetic) <= 1
        assert 0 < e["check"]["max_err_over_quantum"] <= 1.0 and e["check"]["max_rel_grad_err"] < 5e-3, (key, e["check"])
        assert ("roofline" in e) or ("mfma_roofline" in e) or key.startswith("rnntloss_"), key
    assert rec["other_workloads_all_checks_passed"]
    assert other["c5_full_1024_on_one_gpu"]["ms_per_step"] > 4 * other["c5_per_gpu"]["ms_per_step"]
