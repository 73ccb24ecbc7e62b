"""-m gpu: labels that EQUAL the blank symbol -- the one reference edge case where its two locations disagree.

GPU reference (include/detail/gpu_rnnt_kernel.h:161-174): the blank correction and the label correction are independent
`if`s, so at a cell whose label is the blank BOTH are subtracted from the blank column -- the true derivative, since
log p(blank | t,u) then feeds the blank transition and the label transition.  CPU reference
(include/detail/cpu_rnnt.h:253-267): the label term is assigned after the blank term and overwrites it.  The GPU location
of this library follows the GPU reference; the CPU oracle therefore cannot be the checker here -- tests/autograd_ref.py
(fp64 torch.autograd through an explicit log-sum-exp lattice, no gradient formula) is.

Every route that applies the two corrections separately: flat and row-form gradient kernels, packed layout, the three
storage types, the additive joint's conditional epilogues, its one-hot MFMA route (vocabularies <= 256) and the variant
that takes the blank column's corrections from row sums (`BS`: U > 48, small vocabulary), in fp32 and bf16 storage.
"""
import numpy as np
import pytest
import torch

from tests.autograd_ref import rnnt_add_autograd, rnnt_autograd

pytestmark = pytest.mark.gpu

SHAPES = [(2, 9, 5, 7), (2, 40, 70, 50), (1, 20, 9, 1000)]          # N, T, U, A
WHERE = ["first", "last", "middle"]


def blank_of(A, where):
    return {"first": 0, "last": A - 1, "middle": A // 2}[where]


def problem(shape, where, seed, frac=0.4):
    """Random batch with ragged lengths in which >= 30 % of the labels (of every sample's used prefix) are the blank."""
    N, T, U, A = shape
    rng = np.random.default_rng(seed)
    blank = blank_of(A, where)
    labels = rng.integers(0, A, size=(N, U - 1)).astype(np.int32)
    labels[rng.random((N, U - 1)) < frac] = blank
    tl = rng.integers(max(1, T // 2), T + 1, size=N).astype(np.int32); tl[0] = T
    ll = rng.integers((U - 1) // 2, U, size=N).astype(np.int32); ll[-1] = U - 1
    if N == 1:
        tl[0], ll[0] = T, U - 1
    for b in range(N):                                   # at least 30 % of what the sample really uses, incl. its first and last label
        if ll[b] > 0:
            labels[b, 0] = blank
            labels[b, ll[b] - 1] = blank
            k = int(np.ceil(0.3 * ll[b])) - int((labels[b, :ll[b]] == blank).sum())
            if k > 0:
                free = np.flatnonzero(labels[b, :ll[b]] != blank)
                labels[b, rng.choice(free, size=k, replace=False)] = blank
            assert (labels[b, :ll[b]] == blank).mean() >= 0.3
    return labels, tl, ll, blank


def c_abi(x, labels, tl, ll, blank):
    """compute_rnnt_loss / _fp64 / _bf16 through the extension module's gpu_rnnt (the reference binding's signature)."""
    from warprnnt_pytorch import warp_rnnt
    dev = x.device
    costs = torch.zeros(x.shape[0], dtype=x.dtype if x.dtype == torch.float64 else torch.float32)
    grads = torch.full_like(x, 123.0)
    assert warp_rnnt.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev), torch.tensor(ll, device=dev),
                              costs, grads, blank, 0) == 0
    torch.cuda.synchronize()
    return costs.double().numpy(), grads.double().cpu().numpy()


# (cost relative, gradient absolute, gradient relative to the stored value): bf16 = one rounding of the stored element
# (2^-8 |ref|) on top of the fp32 figure -- not an absolute 4e-3, which entries of this size never reach
TOL = {torch.float32: (1e-4, 1e-4, 0.0), torch.float64: (1e-10, 1e-9, 0.0), torch.bfloat16: (1e-4, 1e-4, 2.0 ** -8)}


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_materialised_padded(shape, dtype, where):
    labels, tl, ll, blank = problem(shape, where, 101 + sum(shape))
    rng = np.random.default_rng(7 + sum(shape))
    x = torch.tensor(rng.standard_normal(shape) * 2.0, device="cuda:0").to(dtype).contiguous()
    ref_c, ref_g = rnnt_autograd(x.double().cpu().numpy(), labels, tl, ll, blank)       # on the storage-rounded inputs
    costs, grads = c_abi(x, labels, tl, ll, blank)
    tol_c, tol_g, tol_q = TOL[dtype]
    assert np.abs(costs - ref_c).max() <= tol_c * max(1.0, np.abs(ref_c).max())
    err = np.abs(grads - ref_g) / (tol_g + tol_q * np.abs(ref_g))
    assert err.max() <= 1.0, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # the cells under test carry real mass in the blank column, and the CPU reference's "assignment" answer would fail here
    N = shape[0]
    both = [(b, u) for b in range(N) for u in range(ll[b]) if labels[b, u] == blank]
    assert both and max(abs(ref_g[b, :tl[b], u, blank]).max() for b, u in both) > 1e-2
    for b in range(N):
        assert not grads[b, tl[b]:].any() and not grads[b, :, ll[b] + 1:].any()


@pytest.mark.parametrize("where", WHERE)
def test_materialised_row_form_gradient_kernel(where):
    """acts and grads at different 16-byte phases: grad_rows_kernel instead of the flat stream."""
    from warprnnt_pytorch import warp_rnnt
    shape = (2, 9, 5, 64)
    N, T, U, A = shape
    labels, tl, ll, blank = problem(shape, where, 5)
    dev = torch.device("cuda:0")
    acts = np.random.default_rng(3).standard_normal(shape).astype(np.float32) * 2
    x = torch.zeros(acts.size + 1, device=dev)[1:].view(shape)
    x.copy_(torch.tensor(acts))
    g = torch.zeros(acts.size + 3, device=dev)[3:].view(shape)
    costs = torch.zeros(N)
    assert warp_rnnt.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev), torch.tensor(ll, device=dev),
                              costs, g, blank, 0) == 0
    ref_c, ref_g = rnnt_autograd(acts, labels, tl, ll, blank)
    assert np.abs(costs.numpy() - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    assert np.abs(g.cpu().numpy() - ref_g).max() <= 1e-4


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_materialised_packed(shape, dtype, where):
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
    labels, tl, ll, blank = problem(shape, where, 211 + sum(shape))
    N = shape[0]
    rng = np.random.default_rng(9 + sum(shape))
    dev = torch.device("cuda:0")
    x = torch.tensor(rng.standard_normal(shape) * 2.0, device=dev).to(dtype)
    weights = np.linspace(-1.5, 2.0, N) if N > 1 else np.array([0.75])
    t_lab, t_tl, t_ll = (torch.tensor(v, device=dev) for v in (labels, tl, ll))
    p = pack_joint(x, t_tl, t_ll).contiguous().requires_grad_(True)
    loss = RNNTLossPacked(blank=blank, reduction="none")(p, t_lab, t_tl, t_ll)
    loss.backward(torch.tensor(weights, device=dev, dtype=loss.dtype))
    ref_c, ref_g = rnnt_autograd(x.double().cpu().numpy(), labels, tl, ll, blank, weights)
    ref_pk = np.concatenate([ref_g[b, :tl[b], :ll[b] + 1].reshape(-1, shape[3]) for b in range(N)])
    tol_c, tol_g, tol_q = TOL[dtype]
    assert np.abs(loss.detach().double().cpu().numpy() - ref_c).max() <= tol_c * max(1.0, np.abs(ref_c).max())
    assert (np.abs(p.grad.double().cpu().numpy() - ref_pk) <= tol_g * 2.0 + tol_q * np.abs(ref_pk)).all()   # |weights| <= 2


# ------------------------------------------------------------------------------------------------ additive joint
ADD_SHAPES = SHAPES + [(3, 70, 66, 50), (2, 40, 130, 200), (2, 33, 70, 131), (2, 65, 34, 56), (2, 33, 21, 257)]


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", ADD_SHAPES)
def test_additive_joint(shape, dtype, where):
    """(3,70,66,50) / (2,40,70,50): tiled coefficient kernel + one-hot DF with the blank column from the row sums (`BS`);
    (2,40,130,200): one-hot, two column groups; (2,33,70,131): odd vocabulary, one column per lane; (2,65,34,56): one-hot behind
    the cell-per-thread coefficient kernel (CB plane); (2,33,21,257) and A = 1000: the conditional epilogues."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    N, T, U, A = shape
    labels, tl, ll, blank = problem(shape, where, 307 + sum(shape))
    rng = np.random.default_rng(13 + sum(shape))
    dev = torch.device("cuda:0")
    tf = torch.tensor((rng.standard_normal((N, T, A)) * 1.5).astype(np.float32), device=dev).to(dtype).requires_grad_(True)
    tg = torch.tensor((rng.standard_normal((N, U, A)) * 1.5).astype(np.float32), device=dev).to(dtype).requires_grad_(True)
    loss = RNNTLossAdd(blank=blank, reduction="none")(tf, tg, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev),
                                                      torch.tensor(ll, device=dev))
    loss.sum().backward()
    ref_c, rdf, rdg = rnnt_add_autograd(tf.detach().double().cpu().numpy(), tg.detach().double().cpu().numpy(), labels, tl, ll, blank)
    assert np.abs(loss.detach().double().cpu().numpy() - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    ulp = 5e-5 if dtype == torch.float32 else 2.0 ** -8          # fp32 accumulation of up to T terms / half an ulp of the stored bf16
    edf = np.abs(tf.grad.double().cpu().numpy() - rdf) - (2e-4 * max(1.0, U / 32) + ulp * np.abs(rdf) + 1e-6)
    edg = np.abs(tg.grad.double().cpu().numpy() - rdg) - (2e-4 * max(1.0, T / 32) + ulp * np.abs(rdg) + 1e-6)
    assert edf.max() <= 0, ("df", edf.max(), np.unravel_index(edf.argmax(), edf.shape), blank)
    assert edg.max() <= 0, ("dg", edg.max(), np.unravel_index(edg.argmax(), edg.shape), blank)
    # label rows whose label is the blank: dg's blank column there holds both corrections' mass
    rows = [(b, u) for b in range(N) for u in range(ll[b]) if labels[b, u] == blank]
    assert rows and max(abs(rdg[b, u, blank]) for b, u in rows) > 1e-2
