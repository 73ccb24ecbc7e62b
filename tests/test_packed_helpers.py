"""Host side of the packed layout (warprnnt_pytorch.packed): the row arithmetic the device kernels mirror
(row (b, t, u) = offsets[b] + t * U_b + u).  No GPU."""
import numpy as np
import pytest
import torch

from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint, row_offsets, unpack_joint


def test_offsets_pack_unpack_round_trip():
    rng = np.random.default_rng(0)
    N, T, U, V = 5, 7, 4, 3
    acts = torch.tensor(rng.standard_normal((N, T, U, V)).astype(np.float32))
    tl = torch.tensor([7, 1, 4, 7, 2], dtype=torch.int32)
    ll = torch.tensor([3, 0, 2, 1, 3], dtype=torch.int32)
    offs = row_offsets(tl, ll)
    assert offs.dtype == torch.int64 and offs.tolist() == [0, 28, 29, 41, 55, 63]
    p = pack_joint(acts, tl, ll)
    assert p.shape == (63, V)
    for b in range(N):
        for t in (0, int(tl[b]) - 1):
            for u in (0, int(ll[b])):
                assert torch.equal(p[int(offs[b]) + t * (int(ll[b]) + 1) + u], acts[b, t, u])
    back = unpack_joint(p, tl, ll, T, U)
    for b in range(N):
        assert torch.equal(back[b, :tl[b], :ll[b] + 1], acts[b, :tl[b], :ll[b] + 1])
        assert back[b, tl[b]:].abs().sum() == 0 and back[b, :, ll[b] + 1:].abs().sum() == 0


def test_packed_loss_on_cpu_tensors_equals_padded_loss():
    """CPU tensors: RNNTLossPacked goes through compute_rnnt_loss_packed with RNNT_CPU and must agree with RNNTLoss
    on the padded tensor -- losses and gradients (chain rule through log_softmax included), every reduction."""
    from warprnnt_pytorch import RNNTLoss
    rng = np.random.default_rng(3)
    N, T, U, V = 4, 8, 5, 7
    acts = torch.tensor(rng.standard_normal((N, T, U, V)), dtype=torch.float64)
    labels = torch.tensor(rng.integers(1, V, size=(N, U - 1)), dtype=torch.int32)
    tl = torch.tensor([8, 3, 6, 1], dtype=torch.int32)
    ll = torch.tensor([4, 0, 2, 3], dtype=torch.int32)
    for reduction, w in (("none", torch.tensor([1.0, -2.0, 0.5, 3.0], dtype=torch.float64)), ("mean", None), ("sum", None)):
        a = acts.clone().requires_grad_(True)
        lp = RNNTLoss(reduction=reduction)(a, labels, tl, ll)
        (lp * w).sum().backward() if w is not None else lp.sum().backward()
        p = pack_joint(acts, tl, ll).clone().requires_grad_(True)
        lk = RNNTLossPacked(reduction=reduction)(p, labels, tl, ll)
        (lk * w).sum().backward() if w is not None else lk.sum().backward()
        assert torch.allclose(lk, lp, rtol=1e-12)
        assert torch.allclose(p.grad, pack_joint(a.grad, tl, ll), rtol=1e-10, atol=1e-12)


def test_packed_loss_checks_types_and_row_count():
    acts = torch.zeros((6, 3))
    lab = torch.zeros((1, 1), dtype=torch.int32)
    one = torch.tensor([3], dtype=torch.int32)
    with pytest.raises(TypeError):
        RNNTLossPacked()(acts, lab.long(), one, one)
    with pytest.raises(ValueError):
        RNNTLossPacked()(acts[:5].contiguous(), lab, one, torch.tensor([1], dtype=torch.int32))   # 5 rows, lengths say 6
