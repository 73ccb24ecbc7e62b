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


def test_packed_loss_is_gpu_only_and_checks_types():
    acts = torch.zeros((6, 3))
    lab = torch.zeros((1, 1), dtype=torch.int32)
    one = torch.tensor([3], dtype=torch.int32)
    with pytest.raises(ValueError):
        RNNTLossPacked()(acts, lab, one, torch.tensor([1], dtype=torch.int32))
    with pytest.raises(TypeError):
        RNNTLossPacked()(acts, lab.long(), one, one)
