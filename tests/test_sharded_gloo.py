"""The batch-sharded path with two ranks over gloo (CPU): every rank runs the hot path on its
slab, ONE collective carries the summed loss; result and gradients equal the unsharded run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, reduction, q, ragged=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "warp-transducer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from warprnnt_pytorch.sharded import ShardedRNNTLoss
    acts, labels, tl, ll = _batch()
    n = acts.shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    if ragged:                                  # rank 0 gets three of the four samples, rank 1 the last one
        sl = slice(0, 3) if rank == 0 else slice(3, 4)
    x = acts[sl].clone().requires_grad_(True)
    loss = ShardedRNNTLoss(blank=0, reduction=reduction)(x, labels[sl].contiguous(), tl[sl].contiguous(),
                                                        ll[sl].contiguous())
    w = torch.arange(1, loss.numel() + 1, dtype=loss.dtype)
    (loss * w).sum().backward()
    q.put((rank, loss.detach().numpy(), x.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _batch():
    g = torch.Generator().manual_seed(7)
    N, T, U, A = 4, 6, 4, 5
    acts = torch.randn(N, T, U, A, generator=g)
    labels = torch.randint(1, A, (N, U - 1), generator=g, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32)        # every shard must contain the max lengths
    ll = torch.full((N,), U - 1, dtype=torch.int32)
    return acts, labels, tl, ll


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_two_rank_shard_equals_single(reduction):
    from warprnnt_pytorch import RNNTLoss
    acts, labels, tl, ll = _batch()
    x = acts.clone().requires_grad_(True)
    ref = RNNTLoss(reduction=reduction)(x, labels, tl, ll)
    w = torch.arange(1, ref.numel() + 1, dtype=ref.dtype)
    (ref * w).sum().backward()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduction, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, loss, grad in got:
        assert np.allclose(loss, ref.detach().numpy(), atol=1e-5)          # global loss on every rank
        assert np.allclose(grad, x.grad[rank * 2:(rank + 1) * 2].numpy(), atol=1e-6)


@pytest.mark.parametrize("reduction", ["mean", "none"])
def test_ragged_shards(reduction):
    """A last shard with fewer samples: 'mean' divides by the global count, 'none' returns all costs in batch order."""
    from warprnnt_pytorch import RNNTLoss
    acts, labels, tl, ll = _batch()
    x = acts.clone().requires_grad_(True)
    ref = RNNTLoss(reduction=reduction)(x, labels, tl, ll)
    w = torch.arange(1, ref.numel() + 1, dtype=ref.dtype)
    (ref * w).sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduction, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bounds = [(0, 3), (3, 4)]
    for rank, loss, grad in got:
        assert np.allclose(loss, ref.detach().numpy(), atol=1e-5)
        lo, hi = bounds[rank]
        assert np.allclose(grad, x.grad[lo:hi].numpy(), atol=1e-6)
