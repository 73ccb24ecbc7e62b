"""-m gpu: a whole loss step -- forward AND backward through autograd -- captured in a HIP graph and replayed.

The C entry points only enqueue (tests/test_gpu_parity.py::test_async_entry_is_graph_capturable); this is the same property one
level up, where a training loop needs it: `RNNTLoss(validate=False)` / `RNNTLossAdd(validate=False)` under `torch.cuda.graph`
(workspace, costs and gradients from the caching allocator's capture pool; no device-to-host read, no synchronisation; the
backward node on the capturing stream), replayed on changed activations, against the oracle.  Both bindings."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _capture(step, warmups=3):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmups):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    return graph, out


@pytest.mark.parametrize("loader", ["ext", "ctypes"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rnnt_loss_step_in_a_graph(monkeypatch, oracle, loader, dtype):
    from warprnnt_pytorch import RNNTLoss, warp_rnnt
    if loader == "ctypes":
        monkeypatch.setattr(warp_rnnt, "_EXT", None)
    assert warp_rnnt.binding() == loader
    rng = np.random.default_rng(5)
    N, T, U, A = 3, 21, 9, 40
    blank = 2
    labels = rng.integers(3, A, size=(N, U - 1)).astype(np.int32)
    tl = np.array([T, 13, 17], dtype=np.int32)
    ll = np.array([5, U - 1, 0], dtype=np.int32)
    dev = torch.device("cuda:0")
    x = torch.zeros((N, T, U, A), dtype=dtype, device=dev, requires_grad=True)
    lab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    crit = RNNTLoss(blank=blank, reduction="mean", validate=False)

    def step():
        x.grad = None
        loss = crit(x, lab, ttl, tll)
        loss.backward()
        return loss

    with torch.no_grad():
        x.copy_(torch.tensor(rng.standard_normal((N, T, U, A)), dtype=dtype))
    graph, loss = _capture(step)
    grad = x.grad                                        # the tensor the captured backward writes
    for seed in (1, 2):
        acts = np.random.default_rng(seed).standard_normal((N, T, U, A)).astype(np.float32)
        with torch.no_grad():
            x.copy_(torch.tensor(acts, dtype=dtype))
        graph.replay()
        torch.cuda.synchronize()
        rounded = x.detach().float().cpu().numpy().astype(np.float64)
        ref_c, ref_g, mag = oracle.rnnt_logits(rounded, labels, tl, ll, blank, want_mag=True)
        assert abs(loss.item() - ref_c.mean()) <= 1e-4 * abs(ref_c.mean())
        if dtype == torch.float32:
            assert np.abs(grad.float().cpu().numpy() - ref_g / N).max() <= 1e-4
        oracle.assert_grads(grad.float().cpu().numpy(), ref_g / N, mag / N, dtype, scale=1.0 / N)   # per element, one rounding


@pytest.mark.parametrize("loader", ["ext", "ctypes"])
def test_additive_joint_step_in_a_graph(monkeypatch, oracle, loader):
    from warprnnt_pytorch import warp_rnnt
    from warprnnt_pytorch.add_network import RNNTLossAdd
    if loader == "ctypes":
        monkeypatch.setattr(warp_rnnt, "_EXT", None)
    rng = np.random.default_rng(9)
    N, T, U, A = 2, 30, 70, 50                           # tiled coefficient kernel, one-hot DF with the row sums
    blank = 0
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl = np.array([T, 19], dtype=np.int32)
    ll = np.array([40, U - 1], dtype=np.int32)
    dev = torch.device("cuda:0")
    f = torch.zeros((N, T, A), device=dev, requires_grad=True)
    g = torch.zeros((N, U, A), device=dev, requires_grad=True)
    lab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    crit = RNNTLossAdd(blank=blank, reduction="sum", validate=False)

    def step():
        f.grad = None; g.grad = None
        loss = crit(f, g, lab, ttl, tll)
        loss.backward()
        return loss

    with torch.no_grad():
        f.copy_(torch.tensor(rng.standard_normal((N, T, A)), dtype=torch.float32))
        g.copy_(torch.tensor(rng.standard_normal((N, U, A)), dtype=torch.float32))
    graph, loss = _capture(step)
    df, dg = f.grad, g.grad
    for seed in (3, 4):
        r = np.random.default_rng(seed)
        fa, ga = r.standard_normal((N, T, A)).astype(np.float32), r.standard_normal((N, U, A)).astype(np.float32)
        with torch.no_grad():
            f.copy_(torch.tensor(fa)); g.copy_(torch.tensor(ga))
        graph.replay()
        torch.cuda.synchronize()
        z = fa[:, :, None, :].astype(np.float64) + ga[:, None, :, :].astype(np.float64)
        ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
        assert abs(loss.item() - ref_c.sum()) <= 1e-4 * abs(ref_c.sum())
        assert np.allclose(df.cpu().numpy(), ref_gz.sum(axis=2), rtol=2e-4, atol=5e-4)
        assert np.allclose(dg.cpu().numpy(), ref_gz.sum(axis=1), rtol=2e-4, atol=5e-4)
