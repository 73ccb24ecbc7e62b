"""-m gpu, >= 2 devices: the batch-sharded path with TWO ranks over RCCL (backend "nccl" on ROCm), one process per
GPU -- the twin of tests/test_sharded_gloo.py on the real transport.  Every rank runs the HIP hot path on its slab
(compute_rnnt_loss_fwd / _bwd), ONE all-reduce (or all-gather for 'none') carries the loss; losses and gradients
must equal the unsharded single-GPU run.  Skipped on a 1-GPU box (RCCL refuses two ranks on one device).
Also: `bench.py --gpus 2` launches its own two ranks and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(dev):
    g = torch.Generator().manual_seed(11)
    N, T, U, A = 6, 23, 9, 130
    acts = torch.randn(N, T, U, A, generator=g)
    labels = torch.randint(1, A, (N, U - 1), generator=g, dtype=torch.int32)
    tl = torch.tensor([T, 9, T, T, 14, T], dtype=torch.int32)      # every shard contains the maxima
    ll = torch.tensor([U - 1, 3, 0, U - 1, 5, 2], dtype=torch.int32)
    return [t.to(dev) for t in (acts, labels, tl, ll)]


def _worker(rank, world, port, reduction, q, backend="nccl"):
    for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:                                   # both ranks on device 0, collectives over gloo (staged through the host)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from warprnnt_pytorch.sharded import ShardedRNNTLoss
    acts, labels, tl, ll = _batch(dev)
    n = acts.shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    x = acts[sl].clone().requires_grad_(True)
    loss = ShardedRNNTLoss(blank=0, reduction=reduction)(x, labels[sl].contiguous(), tl[sl].contiguous(),
                                                        ll[sl].contiguous())
    w = torch.arange(1, loss.numel() + 1, dtype=loss.dtype, device=dev)
    (loss * w).sum().backward()
    torch.cuda.synchronize()
    q.put((rank, loss.detach().cpu().numpy(), x.grad.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_two_rccl_ranks_equal_single_gpu(reduction):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one RCCL rank per device)")
    import torch.multiprocessing as mp
    from warprnnt_pytorch import RNNTLoss
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    x = acts.clone().requires_grad_(True)
    ref = RNNTLoss(reduction=reduction)(x, labels, tl, ll)
    w = torch.arange(1, ref.numel() + 1, dtype=ref.dtype, device=dev)
    (ref * w).sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduction, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = acts.shape[0] // 2
    for rank, loss, grad in got:
        assert np.allclose(loss, ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)     # global loss on every rank
        assert np.allclose(grad, x.grad[rank * n:(rank + 1) * n].cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_two_ranks_sharing_one_gpu_over_gloo(reduction):
    """What a 1-GPU box can exercise of the sharded DEVICE path at world_size 2: two processes, both on cuda:0, each
    running the HIP hot path (compute_rnnt_loss_fwd / _bwd) on its shard of the batch, the one collective over gloo
    on the device tensors (RCCL refuses two ranks on one device).  Losses and gradients equal the unsharded run."""
    import torch.multiprocessing as mp
    from warprnnt_pytorch import RNNTLoss
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    x = acts.clone().requires_grad_(True)
    ref = RNNTLoss(reduction=reduction)(x, labels, tl, ll)
    w = torch.arange(1, ref.numel() + 1, dtype=ref.dtype, device=dev)
    (ref * w).sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduction, q, "gloo")) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = acts.shape[0] // 2
    for rank, loss, grad in got:
        assert np.allclose(loss, ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert np.allclose(grad, x.grad[rank * n:(rank + 1) * n].cpu().numpy(), atol=1e-6)


def test_bench_self_launches_two_ranks():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256 and line["dtype"] == "bf16"


def test_bench_sharded_step_on_one_rank():
    """The step `bench.py --gpus N` times for N > 1 (loss + gradients -> RCCL all-reduce of [loss sum, count]; and its
    two-phase variant with the collective beside the gradient pass), on a one-rank RCCL group so that it runs on a one-GPU
    box too: same loss sum as the plain one-GPU step on the same inputs, a complete JSON line as the LAST line of stdout."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c5", "--steps", "4", "--warmup", "1",
            "--no-cpu-baseline", "--no-traffic-pass"]
    lines = []
    for extra in (["--force-sharded"], ["--force-sharded", "--overlap-collective"], []):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
    sh, ov, plain = lines
    assert abs(ov["check"]["loss_sum"] - plain["check"]["loss_sum"]) <= 1e-6 * abs(plain["check"]["loss_sum"])
    assert ov["stage_ms"]["grad"] > 0 and ov["stage_ms"]["row_stats"] > 0      # two calls, one profiled step
    assert sh["n_gpus"] == 1 and sh["scaling"] == "weak" and sh["dtype"] == "bf16" and sh["value"] > 0
    assert "all-reduce" in sh["config"]["parallelism"]
    assert abs(sh["check"]["loss_sum"] - plain["check"]["loss_sum"]) <= 1e-6 * abs(plain["check"]["loss_sum"])
    assert sh["stage_ms"]["grad"] > 0 and sh["stage_ms"]["row_stats"] > 0
    # the multi-GPU line is self-sufficient: ranks RCCL saw, per-rank times, the same workload without the collective
    for line in (sh, ov):
        m = line["multi_gpu"]
        assert m["ranks_seen"] == 1 and m["backend"] == "nccl" and len(m["per_rank_ms"]) == 1
        assert m["single_gpu_same_workload_ms"] > 0 and 0.3 < m["scaling_efficiency"] <= 1.5
        assert abs(m["per_rank_ms"][0] - line["value"]) <= 1e-3 * line["value"]
    assert "multi_gpu" not in plain
    # every line carries its own parity evidence (two samples of the timed batch against the fp64 oracle)
    for line in lines:
        assert line["check"]["passed"] and line["check"]["max_abs_grad_err"] <= 4e-3, line["check"]


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus N` with fewer than N devices must fail loudly, never fall back to fewer GPUs."""
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 2
    assert "refusing" in out.stderr and not out.stdout.strip()
