"""Packed ("compact") activations (include/rnnt.h compute_rnnt_loss_packed*, warprnnt_pytorch.packed): sample b
is only its T_b x U_b rows.  Checked against the padded path of this library (itself pinned to the oracle in
test_gpu_parity.py) on the same problem, and against the fp64 CPU oracle directly."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # N, T, U, A
    (3, 9, 5, 7), (2, 20, 9, 40), (4, 6, 3, 5000), (2, 33, 70, 50), (1, 1, 1, 9), (3, 5, 1, 33),
    (6, 3, 2, 7),           # the whole batch inside one chunk of the gradient kernel: six samples per block
    (2, 12, 6, 3100),       # 12.4 KB rows: block-per-row statistics kernel
    (2, 10, 4, 512), (5, 17, 8, 130),
]


def problem(shape, seed, dtype=np.float32):
    N, T, U, A = shape
    rng = np.random.default_rng(seed)
    acts = (rng.standard_normal((N, T, U, A)) * 2).astype(dtype)
    blank = int(rng.integers(0, A))
    labels = rng.integers(0, A, size=(N, U - 1)).astype(np.int32)
    labels[labels == blank] = (blank + 1) % A
    tl = rng.integers(1, T + 1, size=N).astype(np.int32)
    ll = rng.integers(0, U, size=N).astype(np.int32)
    tl[rng.integers(0, N)] = T
    ll[rng.integers(0, N)] = U - 1
    return acts, labels, tl, ll, blank


def run_both(acts, labels, tl, ll, blank, weights, fastemit=0.0, dtype=torch.float32):
    from warprnnt_pytorch import RNNTLoss
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
    dev = torch.device("cuda:0")
    t_lab, t_tl, t_ll = (torch.tensor(x, device=dev) for x in (labels, tl, ll))
    w = torch.tensor(weights, device=dev)
    a = torch.tensor(acts, device=dev).to(dtype).requires_grad_(True)
    loss = RNNTLoss(blank=blank, reduction="none", fastemit_lambda=fastemit)(a, t_lab, t_tl, t_ll)
    loss.backward(w.to(loss.dtype))
    p = pack_joint(a.detach(), t_tl, t_ll).contiguous().requires_grad_(True)
    lp = RNNTLossPacked(blank=blank, reduction="none", fastemit_lambda=fastemit)(p, t_lab, t_tl, t_ll)
    lp.backward(w.to(lp.dtype))
    return (loss.detach().double().cpu().numpy(), pack_joint(a.grad, t_tl, t_ll).double().cpu().numpy(),
            lp.detach().double().cpu().numpy(), p.grad.double().cpu().numpy())


@pytest.mark.parametrize("shape", SHAPES)
def test_packed_equals_padded_and_oracle(oracle, shape):
    acts, labels, tl, ll, blank = problem(shape, sum(shape))
    N = shape[0]
    weights = np.linspace(-1.5, 2.5, N).astype(np.float32) if N > 1 else np.array([0.75], dtype=np.float32)
    c_pad, g_pad, c_pk, g_pk = run_both(acts, labels, tl, ll, blank, weights)
    assert g_pk.shape == (int((tl.astype(np.int64) * (ll + 1)).sum()), shape[3])
    # the two layouts take different statistics kernels for short rows (LDS tiles are a padded-layout form): the
    # summation order of log Z differs, and the lattice carries that fp32 noise -- the oracle below is the bar
    assert np.allclose(c_pk, c_pad, rtol=1e-5, atol=1e-5)
    assert np.allclose(g_pk, g_pad, rtol=1e-4, atol=1e-5)
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll, blank)
    assert np.abs(c_pk - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    ref_pk = np.concatenate([(ref_g[b, :tl[b], :ll[b] + 1] * weights[b]).reshape(-1, shape[3]) for b in range(N)])
    assert np.abs(g_pk - ref_pk).max() <= 1e-3 * max(1.0, np.abs(ref_pk).max())


def test_mean_reduction_and_host_maxima():
    from warprnnt_pytorch import RNNTLoss
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
    acts, labels, tl, ll, blank = problem((4, 11, 6, 37), 5)
    dev = torch.device("cuda:0")
    t_lab, t_tl, t_ll = (torch.tensor(x, device=dev) for x in (labels, tl, ll))
    a = torch.tensor(acts, device=dev, requires_grad=True)
    RNNTLoss(blank=blank)(a, t_lab, t_tl, t_ll).backward()
    p = pack_joint(a.detach(), t_tl, t_ll).requires_grad_(True)
    loss = RNNTLossPacked(blank=blank)(p, t_lab, t_tl, t_ll, max_T=11, max_U=6)      # no host round trip
    loss.backward()
    assert torch.allclose(p.grad, pack_joint(a.grad, t_tl, t_ll), rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        RNNTLossPacked(blank=blank)(p.detach()[:-1].contiguous(), t_lab, t_tl, t_ll)   # row count disagrees with the lengths


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float64])
def test_storage_types(dtype):
    acts, labels, tl, ll, blank = problem((3, 10, 4, 512), 21)
    weights = np.array([1.0, -0.5, 2.0], dtype=np.float32)
    c_pad, g_pad, c_pk, g_pk = run_both(acts, labels, tl, ll, blank, weights, dtype=dtype)
    tol = 1e-9 if dtype is torch.float64 else 1e-6
    assert np.allclose(c_pk, c_pad, rtol=max(tol, 1e-5), atol=1e-5)
    # both layouts compute in fp32 and round once on store: they may differ by ONE ulp of the stored value, not by an
    # absolute 1e-2 (which no entry of a 512-symbol row reaches)
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float64: 1e-9}[dtype]
    assert np.allclose(g_pk, g_pad, rtol=ulp + 1e-4, atol=4e-6 if dtype is not torch.float64 else 1e-12)


def test_fastemit_on_packed_activations():
    acts, labels, tl, ll, blank = problem((3, 14, 7, 60), 8)
    weights = np.array([1.0, 1.0, 1.0], dtype=np.float32)
    c_pad, g_pad, c_pk, g_pk = run_both(acts, labels, tl, ll, blank, weights, fastemit=0.01)
    assert np.allclose(c_pk, c_pad, rtol=1e-5) and np.allclose(g_pk, g_pad, rtol=1e-4, atol=1e-5)
    _, g0, _, _ = run_both(acts, labels, tl, ll, blank, weights)
    assert np.abs(g_pad - g0).max() > 1e-4          # the regulariser is active


def test_single_call_entry_and_validation():
    from warprnnt_pytorch import _lib
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint, row_offsets
    acts, labels, tl, ll, blank = problem((3, 8, 5, 7), 3)
    N, T, U, A = acts.shape
    dev = torch.device("cuda:0")
    t_lab, t_tl, t_ll = (torch.tensor(x, device=dev) for x in (labels, tl, ll))
    p = pack_joint(torch.tensor(acts, device=dev), t_tl, t_ll).contiguous()
    R = p.shape[0]
    offs = row_offsets(t_tl, t_ll)
    lib = _lib.lib()
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    costs = torch.empty(N, device=dev)
    grads = torch.full_like(p, 7.0)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=blank,
                           maxT=T, maxU=U, batch_first=True)

    def call(acts_ptr=p.data_ptr(), grads_ptr=grads.data_ptr(), rows=R, offs_ptr=offs.data_ptr()):
        return lib.compute_rnnt_loss_packed(acts_ptr, grads_ptr, t_lab.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(),
                                            offs_ptr, rows, A, N, costs.data_ptr(), None, ws.data_ptr(), opt, 0, 0.0)

    assert call() == 0
    torch.cuda.synchronize()
    q = p.clone().requires_grad_(True)
    loss = RNNTLossPacked(blank=blank, reduction="none")(q, t_lab, t_tl, t_ll)
    loss.sum().backward()
    assert torch.allclose(costs, loss.detach(), rtol=1e-6) and torch.allclose(grads, q.grad, rtol=1e-5, atol=2e-6)
    assert call(rows=0) == 2 and call(rows=-5) == 2 and call(rows=N * T * U + 1) == 2 and call(offs_ptr=None) == 2
    assert call(acts_ptr=p.data_ptr() + 4, grads_ptr=grads.data_ptr() + 4, rows=R - 1) == 2   # not 16-byte aligned
    assert lib.compute_rnnt_loss_packed(p.data_ptr(), None, t_lab.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(),
                                        offs.data_ptr(), R, A, N, costs.data_ptr(), None, ws.data_ptr(), opt, 0,
                                        0.0) == 0                                                # score only
    torch.cuda.synchronize()
    assert torch.allclose(costs, loss.detach(), rtol=1e-6)


def test_packed_two_phase_is_graph_capturable(oracle):
    """Forward + backward of the packed entries (incl. the per-row scale expansion) are enqueue-only: captured in
    a HIP graph and replayed on changed activations and scales."""
    from warprnnt_pytorch import _lib
    from warprnnt_pytorch.packed import pack_joint, row_offsets
    acts, labels, tl, ll, blank = problem((3, 12, 6, 40), 17)
    N, T, U, A = acts.shape
    dev = torch.device("cuda:0")
    t_lab, t_tl, t_ll = (torch.tensor(x, device=dev) for x in (labels, tl, ll))
    p = pack_joint(torch.tensor(acts, device=dev), t_tl, t_ll).contiguous()
    R = p.shape[0]
    offs = row_offsets(t_tl, t_ll)
    lib = _lib.lib()
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    costs, grads = torch.zeros(N, device=dev), torch.zeros_like(p)
    scale = torch.ones(N, device=dev)

    def enqueue():
        opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream,
                               blank_label=blank, maxT=T, maxU=U, batch_first=True)
        assert lib.compute_rnnt_loss_packed_fwd(p.data_ptr(), t_lab.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(),
                                                offs.data_ptr(), R, A, N, costs.data_ptr(), ws.data_ptr(), opt, 0, 1,
                                                0.0) == 0
        assert lib.compute_rnnt_loss_packed_bwd(p.data_ptr(), grads.data_ptr(), scale.data_ptr(), offs.data_ptr(), R, A,
                                                N, ws.data_ptr(), opt, 0) == 0

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside capture
        enqueue()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        enqueue()
    for f, w in ((1.0, [1.0, 1.0, 1.0]), (0.5, [2.0, -1.0, 0.25])):
        p.copy_(pack_joint(torch.tensor(acts * f, device=dev), t_tl, t_ll))
        scale.copy_(torch.tensor(w))
        costs.zero_(); grads.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ref_c, ref_g = oracle.rnnt_logits((acts * f).astype(np.float32).astype(np.float64), labels, tl, ll, blank)
        ref_pk = np.concatenate([(ref_g[b, :tl[b], :ll[b] + 1] * w[b]).reshape(-1, A) for b in range(N)])
        assert np.abs(costs.cpu().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
        assert np.abs(grads.cpu().numpy() - ref_pk).max() < 2e-4



def test_supplied_maxima_are_checked_on_the_device(oracle):
    """max_T / max_U supplied by the caller (no host round trip): a row count that disagrees with the lengths, or maxima
    smaller than the batch's, give NaN losses instead of silent garbage, and nothing is accessed out of bounds (the
    valid samples of a batch with one too-long sample are still right)."""
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31)
    N, T, U, A = 4, 12, 6, 300
    acts = torch.tensor(rng.standard_normal((N, T, U, A)).astype(np.float32), device=dev)
    labels = torch.tensor(rng.integers(1, A, size=(N, U - 1)).astype(np.int32), device=dev)
    tl = torch.tensor([T, 7, 9, T], dtype=torch.int32, device=dev)
    ll = torch.tensor([U - 1, 2, 4, 3], dtype=torch.int32, device=dev)
    packed = pack_joint(acts, tl, ll).contiguous()
    crit = RNNTLossPacked(blank=0, reduction="none")
    good = crit(packed, labels, tl, ll, max_T=T, max_U=U)
    ref_c, _ = oracle.rnnt_logits(acts.double().cpu().numpy(), labels.cpu().numpy(), tl.cpu().numpy(), ll.cpu().numpy())
    assert np.abs(good.cpu().numpy() - ref_c).max() < 1e-4 * np.abs(ref_c).max()
    short = crit(packed[:-5].contiguous(), labels, tl, ll, max_T=T, max_U=U)       # fewer rows than the lengths describe
    assert torch.isnan(short).all()
    small = crit(packed, labels[:, :U - 2].contiguous(), tl, ll, max_T=T - 1, max_U=U - 1)   # maxima below samples 0 and 3
    sc = small.cpu().numpy()
    assert np.isnan(sc[0]) and np.isnan(sc[3]) and not np.isnan(sc[1])
    torch.cuda.synchronize()
