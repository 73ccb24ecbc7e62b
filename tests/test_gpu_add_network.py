"""-m gpu: the additive-joint entry (compute_rnnt_loss_add) against the CPU
oracle run on the MATERIALISED joint  z[b,t,u,:] = f[b,t,:] + g[b,u,:]  -- costs, and
df = sum_u dz, dg = sum_t dz (docs/rnnt_notes.tex:147-153) -- and against this library's own
materialised path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # N, T, U, A
    (2, 7, 4, 5), (3, 20, 9, 40), (2, 33, 21, 257), (1, 50, 130, 12), (2, 9, 300, 7), (1, 70, 3, 1000),
    (2, 40, 16, 64), (1, 1, 1, 9), (2, 5, 1, 33), (2, 1, 6, 31), (3, 65, 33, 100),
    (2, 40, 35, 2048), (1, 34, 5, 5000), (2, 31, 70, 1030),   # vocabulary split over 8 / 4 wavefronts of a tile
    (2, 65, 34, 56), (2, 40, 70, 57), (3, 33, 33, 2), (2, 64, 64, 50),   # either side of the small-vocabulary Z kernel's limit
    (2, 600, 5, 300), (1, 520, 3, 129),    # long utterances: DG splits the contraction over t for any vocabulary (T >= 512)
    (2, 30, 70, 128), (2, 20, 66, 256),    # DF columns per lane follow the vocabulary (A = 128: one, A = 256: two)
]


def problem(shape, seed):
    N, T, U, A = shape
    rng = np.random.default_rng(seed)
    f = (rng.standard_normal((N, T, A)) * 1.5).astype(np.float32)
    g = (rng.standard_normal((N, U, A)) * 1.5).astype(np.float32)
    blank = int(rng.integers(0, A))
    labels = rng.integers(0, A, size=(N, U - 1))
    labels[labels == blank] = (blank + 1) % A
    tl = rng.integers(1, T + 1, size=N); tl[0] = T
    ll = rng.integers(0, U, size=N); ll[-1] = U - 1
    if N == 1:
        tl[0], ll[0] = T, U - 1
    return f, g, labels.astype(np.int32), tl.astype(np.int32), ll.astype(np.int32), blank


def run_add(f, g, labels, tl, ll, blank, reduction="none"):
    from warprnnt_pytorch.add_network import RNNTLossAdd
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev, requires_grad=True)
    tg = torch.tensor(g, device=dev, requires_grad=True)
    lab = torch.tensor(labels, device=dev) if labels.size else torch.zeros((f.shape[0], 0), dtype=torch.int32, device=dev)
    loss = RNNTLossAdd(blank=blank, reduction=reduction)(tf, tg, lab, torch.tensor(tl, device=dev),
                                                        torch.tensor(ll, device=dev))
    loss.sum().backward()
    return loss.detach().cpu().numpy().astype(np.float64), tf.grad.cpu().numpy(), tg.grad.cpu().numpy()


@pytest.mark.parametrize("shape", SHAPES)
def test_against_oracle_on_materialised_joint(oracle, shape):
    f, g, labels, tl, ll, blank = problem(shape, sum(shape))
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    lab = labels if labels.size else np.zeros((shape[0], 1), dtype=np.int32)[:, :0]
    ref_c, ref_gz = oracle.rnnt_logits(z, lab, tl, ll, blank)
    costs, df, dg = run_add(f, g, labels, tl, ll, blank)
    assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    # df sums U per-cell gradients, dg sums T of them: the per-element bound of the materialised
    # path (1e-4 .. 1e-3, north_star) accumulates, so the absolute tolerance scales with the count
    # (the blank column of dg reaches -T: fp32 accumulation adds a relative 5e-5; summing the
    # MATERIALISED GPU path's gradients over t shows the same error)
    N, T, U, A = shape
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    assert (np.abs(df - rdf) <= 2e-4 * max(1.0, U / 32) + 5e-5 * np.abs(rdf)).all()
    assert (np.abs(dg - rdg) <= 2e-4 * max(1.0, T / 32) + 5e-5 * np.abs(rdg)).all()
    for b in range(shape[0]):                      # padded time steps / label positions get zero gradient
        assert not df[b, tl[b]:].any() and not dg[b, ll[b] + 1:].any()


@pytest.mark.parametrize("where", ["first", "last", "middle"])
@pytest.mark.parametrize("shape", [(3, 70, 66, 50), (2, 40, 130, 200), (2, 33, 70, 131)])   # split contraction / 2 column groups / odd rows (one column per lane)
def test_blank_column_from_row_sums(oracle, shape, where):
    """fp32, U > 48, small vocabulary: the coefficient kernel writes NO cb plane; joint_df_kernel<..., BS> takes the blank
    column's corrections from the row sums the coefficient kernel forms (rnnt_joint_kernels.h).  The blank symbol at either
    end of the vocabulary and inside it, ragged lengths: df's blank column against the oracle on the materialised joint."""
    f, g, labels, tl, ll, _ = problem(shape, 31 + sum(shape))
    N, T, U, A = shape
    blank = {"first": 0, "last": A - 1, "middle": A // 2}[where]
    labels = np.where(labels == blank, (blank + 1) % A, labels).astype(np.int32)
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    costs, df, dg = run_add(f, g, labels, tl, ll, blank)
    assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    assert (np.abs(df - rdf) <= 2e-4 * max(1.0, U / 32) + 5e-5 * np.abs(rdf)).all()
    assert (np.abs(dg - rdg) <= 2e-4 * max(1.0, T / 32) + 5e-5 * np.abs(rdg)).all()
    assert np.abs(rdf[..., blank]).max() > 0.5          # the column under test carries real mass
    for b in range(N):
        assert not df[b, tl[b]:].any() and not dg[b, ll[b] + 1:].any()


def test_equals_materialised_gpu_path():
    from warprnnt_pytorch import RNNTLoss
    f, g, labels, tl, ll, blank = problem((4, 30, 12, 300), 5)
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev, requires_grad=True)
    tg = torch.tensor(g, device=dev, requires_grad=True)
    args = [torch.tensor(a, device=dev) for a in (labels, tl, ll)]
    joint = tf.unsqueeze(2) + tg.unsqueeze(1)
    ref = RNNTLoss(blank=blank, reduction="mean")(joint.contiguous(), *args)
    ref.sum().backward()
    costs, df, dg = run_add(f, g, labels, tl, ll, blank, reduction="mean")
    assert np.allclose(costs, ref.item(), rtol=1e-5)
    assert np.allclose(df, tf.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)
    assert np.allclose(dg, tg.grad.cpu().numpy(), rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("sep", [15.0, 30.0, 45.0, 70.0, 100.0])
def test_peak_separation_sweep(oracle, sep):
    """exp(f) exp(g) factorisation: some time rows peak `sep` nats above the rest at one symbol, all
    label rows at another.  Small separations stay in the GEMMs, large ones take the direct branches
    (cells recomputed in the Z kernel, far cells added by the fix-up kernel); both kinds mix here."""
    f, g, labels, tl, ll, blank = problem((2, 37, 12, 96), int(sep))
    f[:, ::3, 5] += sep
    g[..., 60] += sep
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    costs, df, dg = run_add(f, g, labels, tl, ll, blank)
    assert np.isfinite(costs).all() and np.abs(costs - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    # logits of magnitude 2*sep carry fp32 input rounding of ~1e-5 per cell, which the lattice
    # accumulates (costs are in the thousands): the materialised GPU path deviates from the fp64
    # oracle by the same amount (measured: 2e-3 abs on elements of size 6.6 at sep = 70), so the
    # bound is north_star's gradient bar, 1e-3 relative
    assert np.allclose(df, ref_gz.sum(axis=2), rtol=1e-3, atol=5e-4)
    assert np.allclose(dg, ref_gz.sum(axis=1), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("shape", [(1, 6, 4, 50), (1, 6, 70, 50), (1, 5, 1, 20),    # cell / tiled coefficient kernel, U = 1
                                   (3, 40, 70, 50), (3, 33, 20, 300), (2, 20, 70, 300)])   # ragged batches; only sample 1 has far cells; tiled + epilogue corrections
def test_large_logit_range_is_safe(oracle, shape):
    """Rows whose best f column and best g column differ by 100+ nats: every cell is recomputed directly, and the
    gradient GEMMs leave those cells out (W = the far mark, a negative zero; joint_far_kernel adds them from the c the
    coefficient kernel stored for exactly those cells)."""
    f, g, labels, tl, ll, blank = problem(shape, 77)
    far = slice(1, 2) if shape[0] > 1 else slice(None)
    f[far, :, 3] += 120.0
    g[far, :, 40 % shape[3]] += 150.0
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    costs, df, dg = run_add(f, g, labels, tl, ll, blank)
    assert np.isfinite(costs).all() and np.abs(costs - ref_c).max() <= 2e-4 * np.abs(ref_c).max()
    # logits of magnitude 270 carry an fp32 rounding of 3e-5 in every exponent: the peak column's entries
    # (up to 65 here) come out 2e-5 .. 3e-4 relative from the fp64 oracle in every mode of the library
    # (one-hot / atomics, tiled / cell coefficients give the same numbers) -- north_star's gradient bar, 1e-3
    assert np.allclose(df, ref_gz.sum(axis=2), rtol=1e-3, atol=5e-4)
    assert np.allclose(dg, ref_gz.sum(axis=1), rtol=1e-3, atol=5e-4)


@pytest.mark.parametrize("A", [200, 203, 600, 1200])       # 1 / 1 (unaligned rows: scalar loads) / 4 / 8 wavefronts per tile
@pytest.mark.parametrize("bump", [20.0, 70.0, "masked"])
def test_sampled_row_references_and_their_guard(oracle, A, bump):
    """Vocabularies of 64 symbols and more: the Z kernel takes the maximum of a row's FIRST 32 columns as the exponent
    reference instead of a row-maximum pass (rnnt_joint_kernels.h, SAMPLED).  bump = 20: some rows peak 20 nats above
    that sample, inside the guard (40 in base 2 = 27.7 nats) -- the sampled reference is used as it is; 70: beyond the
    guard -- the exact pass behind it takes over; "masked": the first 32 columns are -inf, so there is no finite
    sample at all.  Every case against the fp64 oracle on the materialised joint."""
    N, T, U = 3, 37, 9
    f, g, labels, tl, ll, blank = problem((N, T, U, A), 5 + A)
    blank = 40 + blank % (A - 40)                          # keep blank and labels out of the masked columns
    labels = (40 + labels % (A - 40)).astype(np.int32)
    labels[labels == blank] = 40 + (blank - 39) % (A - 40)
    if bump == "masked":
        f[..., :32] = -np.inf
        g[..., :32] = -np.inf
    else:
        f[0, ::3, 32 + (A // 3)] += bump                   # some rows of some samples, far from the sampled columns
        f[2, 5, A - 1] += bump
        g[1, 2::2, 32 + (A // 2)] += 1.5 * bump
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    costs, df, dg = run_add(f, g, labels, tl, ll, blank)
    assert np.isfinite(costs).all() and np.abs(costs - ref_c).max() <= 2e-4 * max(1.0, np.abs(ref_c).max())
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    assert np.isfinite(df).all() and np.isfinite(dg).all()
    assert np.allclose(df, rdf, rtol=1e-3, atol=5e-4)
    assert np.allclose(dg, rdg, rtol=1e-3, atol=5e-4 * max(1.0, T / 32))


def test_per_sample_grad_output_is_folded_in(oracle):
    """reduction='none' with a non-uniform grad_output: the two-phase backward multiplies sample b's
    gradients by grad_output[b] inside the gradient kernels."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    f, g, labels, tl, ll, blank = problem((3, 21, 7, 130), 11)
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev, requires_grad=True)
    tg = torch.tensor(g, device=dev, requires_grad=True)
    w = np.array([0.5, -2.0, 3.25], dtype=np.float32)
    loss = RNNTLossAdd(blank=blank, reduction="none")(tf, tg, torch.tensor(labels, device=dev),
                                                     torch.tensor(tl, device=dev), torch.tensor(ll, device=dev))
    loss.backward(torch.tensor(w, device=dev))
    assert np.allclose(loss.detach().cpu().numpy(), ref_c, rtol=1e-5)
    assert np.allclose(tf.grad.cpu().numpy(), ref_gz.sum(axis=2) * w[:, None, None], rtol=1e-4, atol=2e-4)
    assert np.allclose(tg.grad.cpu().numpy(), ref_gz.sum(axis=1) * w[:, None, None], rtol=1e-4, atol=5e-4)


def test_single_call_entry_equals_two_phase():
    """compute_rnnt_loss_add (one call, through ctypes) against the module's fwd / bwd route."""
    from warprnnt_pytorch import _lib
    f, g, labels, tl, ll, blank = problem((2, 40, 9, 260), 3)
    costs2, df2, dg2 = run_add(f, g, labels, tl, ll, blank)
    dev = torch.device("cuda:0")
    tf, tg = torch.tensor(f, device=dev), torch.tensor(g, device=dev)
    tlab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    N, T, A = f.shape
    U = g.shape[1]
    df, dg = torch.full_like(tf, 7.0), torch.full_like(tg, 7.0)      # every element must be overwritten
    costs = torch.empty(N, device=dev)
    ws = torch.empty(_lib.workspace_bytes_add(T, U, N), dtype=torch.uint8, device=dev)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=blank,
                           maxT=T, maxU=U, batch_first=True)
    lib = _lib.lib()
    st = lib.compute_rnnt_loss_add(tf.data_ptr(), tg.data_ptr(), df.data_ptr(), dg.data_ptr(), tlab.data_ptr(),
                                   tll.data_ptr(), ttl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
    assert st == 0
    torch.cuda.synchronize()
    assert np.array_equal(costs.cpu().numpy().astype(np.float64), costs2)
    assert np.array_equal(df.cpu().numpy(), df2) and np.array_equal(dg.cpu().numpy(), dg2)
    # score only: NULL gradients, and one NULL gradient is rejected
    st = lib.compute_rnnt_loss_add(tf.data_ptr(), tg.data_ptr(), None, None, tlab.data_ptr(), tll.data_ptr(),
                                   ttl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
    assert st == 0
    torch.cuda.synchronize()
    assert np.array_equal(costs.cpu().numpy().astype(np.float64), costs2)
    st = lib.compute_rnnt_loss_add(tf.data_ptr(), tg.data_ptr(), df.data_ptr(), None, tlab.data_ptr(), tll.data_ptr(),
                                   ttl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
    assert st == 2


def test_graph_capture_and_replay(oracle):
    """compute_rnnt_loss_add only enqueues (no allocation, no host sync): capture it in a hipGraph and
    replay it on changed inputs."""
    from warprnnt_pytorch import _lib
    f, g, labels, tl, ll, blank = problem((3, 25, 8, 96), 21)
    dev = torch.device("cuda:0")
    N, T, A = f.shape
    U = g.shape[1]
    tf, tg = torch.tensor(f, device=dev), torch.tensor(g, device=dev)
    tlab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    df, dg, costs = torch.empty_like(tf), torch.empty_like(tg), torch.empty(N, device=dev)
    ws = torch.empty(_lib.workspace_bytes_add(T, U, N), dtype=torch.uint8, device=dev)
    lib = _lib.lib()

    def call():
        opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream,
                               blank_label=blank, maxT=T, maxU=U, batch_first=True)
        assert lib.compute_rnnt_loss_add(tf.data_ptr(), tg.data_ptr(), df.data_ptr(), dg.data_ptr(), tlab.data_ptr(),
                                         tll.data_ptr(), ttl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt) == 0

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside capture
        call()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for scale in (1.0, 0.5):
        tf.copy_(torch.tensor(f * scale)); tg.copy_(torch.tensor(g * scale))
        costs.zero_(); df.zero_(); dg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        z = (f * scale)[:, :, None, :].astype(np.float64) + (g * scale)[:, None, :, :].astype(np.float64)
        ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
        assert np.abs(costs.cpu().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
        assert np.allclose(df.cpu().numpy(), ref_gz.sum(axis=2), rtol=1e-4, atol=2e-4)
        assert np.allclose(dg.cpu().numpy(), ref_gz.sum(axis=1), rtol=1e-4, atol=5e-4)


def test_masked_vocabulary_entries(oracle):
    """-inf in trans_acts / pred_acts columns that are neither the blank nor a label: probability zero,
    gradient exactly zero there, no NaN anywhere."""
    f, g, labels, tl, ll, blank = problem((2, 19, 6, 64), 8)
    labels = (labels % 20).astype(np.int32)
    blank = 25
    f[..., 40:50] = -np.inf
    g[:, :, 52:60] = -np.inf
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    costs, df, dg = run_add(f, g, labels, tl, ll, blank)
    assert np.isfinite(costs).all() and np.abs(costs - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    assert not np.isnan(df).any() and not np.isnan(dg).any()
    for sl in (slice(40, 50), slice(52, 60)):
        assert not df[..., sl].any() and not dg[..., sl].any()
    assert np.allclose(df, ref_gz.sum(axis=2), rtol=1e-4, atol=2e-4)
    assert np.allclose(dg, ref_gz.sum(axis=1), rtol=1e-4, atol=5e-4)


def test_fastemit_on_the_additive_joint(oracle):
    """fastemit_lambda on RNNTLossAdd: the oracle's label log-prob gradients times (1 + lambda), chain rule,
    then the sums over u / t."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    lam = 0.05
    f, g, labels, tl, ll, blank = problem((3, 23, 9, 130), 14)
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    lp = oracle.log_softmax(z)
    ref_c, g_lp = oracle.rnnt_logprobs(lp, labels, tl, ll, blank)
    for b in range(f.shape[0]):
        for u in range(ll[b]):
            g_lp[b, :tl[b], u, labels[b, u]] *= 1.0 + lam
    ref_gz = oracle.chain_rule_to_logits(lp, g_lp)
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev, requires_grad=True)
    tg = torch.tensor(g, device=dev, requires_grad=True)
    loss = RNNTLossAdd(blank=blank, reduction="none", fastemit_lambda=lam)(
        tf, tg, *(torch.tensor(a, device=dev) for a in (labels, tl, ll)))
    loss.sum().backward()
    assert np.allclose(loss.detach().cpu().numpy(), ref_c, rtol=1e-5)
    assert np.allclose(tf.grad.cpu().numpy(), ref_gz.sum(axis=2), rtol=1e-4, atol=2e-4)
    assert np.allclose(tg.grad.cpu().numpy(), ref_gz.sum(axis=1), rtol=1e-4, atol=5e-4)


def test_validation_errors():
    from warprnnt_pytorch.add_network import rnnt_loss_add
    dev = torch.device("cuda:0")
    f, g = torch.zeros(2, 4, 5, device=dev), torch.zeros(2, 3, 5, device=dev)
    lab = torch.ones(2, 2, dtype=torch.int32, device=dev)
    tl, ll = torch.tensor([4, 4], dtype=torch.int32, device=dev), torch.tensor([2, 2], dtype=torch.int32, device=dev)
    rnnt_loss_add(f, g, lab, tl, ll)
    with pytest.raises(ValueError):
        rnnt_loss_add(f, torch.zeros(2, 3, 6, device=dev), lab, tl, ll)
    with pytest.raises(TypeError):
        rnnt_loss_add(f.double(), g.double(), lab, tl, ll)
    with pytest.raises(ValueError):
        rnnt_loss_add(f.cpu(), g.cpu(), lab.cpu(), tl.cpu(), ll.cpu())
    with pytest.raises(ValueError):
        rnnt_loss_add(f, g, lab, torch.tensor([3, 3], dtype=torch.int32, device=dev), ll)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 20, 9, 40), (2, 33, 21, 257), (2, 40, 35, 2048), (2, 9, 300, 7), (1, 34, 5, 5000),
                                   (2, 40, 70, 57), (2, 65, 34, 56), (1, 1, 1, 9), (2, 31, 70, 1030), (3, 65, 33, 100),
                                   (2, 600, 5, 300), (2, 30, 70, 128), (2, 20, 66, 256)])   # split DG at long T; DF columns per lane by vocabulary
def test_sixteen_bit_activations(oracle, dtype, shape):
    """bfloat16 / float16 STORAGE of both activations and both gradients (compute_rnnt_loss_add_fwd_dt / _bwd_dt: the
    kernels read and write the 16-bit tensors directly and compute in fp32): the loss against the oracle on the
    ROUNDED inputs, the gradients within that dtype's rounding of the exact result.  Shapes: every Z kernel (small
    vocabulary, 1 / 4 / 8 wavefronts per tile), one-hot and conditional label corrections, odd vocabularies
    (scalar loads), wide lattices."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    f, g, labels, tl, ll, blank = problem(shape, sum(shape) + 3)
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev).to(dtype).requires_grad_(True)
    tg = torch.tensor(g, device=dev).to(dtype).requires_grad_(True)
    lab = torch.tensor(labels, device=dev) if labels.size else torch.zeros((f.shape[0], 0), dtype=torch.int32, device=dev)
    args = (lab, torch.tensor(tl, device=dev), torch.tensor(ll, device=dev))
    loss = RNNTLossAdd(blank=blank, reduction="none")(tf, tg, *args)
    loss.sum().backward()
    assert loss.dtype == torch.float32 and tf.grad.dtype == dtype and tg.grad.dtype == dtype
    fr, gr = tf.detach().double().cpu().numpy(), tg.detach().double().cpu().numpy()
    z = fr[:, :, None, :] + gr[:, None, :, :]
    ref_c, ref_gz = oracle.rnnt_logits(z, labels if labels.size else np.zeros((shape[0], 1), dtype=np.int32)[:, :0], tl, ll, blank)
    assert np.abs(loss.detach().double().cpu().numpy() - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11          # half an ulp of the STORED gradient, relative
    N, T, U, A = shape
    assert (np.abs(tf.grad.double().cpu().numpy() - rdf) <= 2e-4 * max(1.0, U / 32) + ulp * np.abs(rdf) + 1e-6).all()
    assert (np.abs(tg.grad.double().cpu().numpy() - rdg) <= 2e-4 * max(1.0, T / 32) + ulp * np.abs(rdg) + 1e-6).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sixteen_bit_far_cells_and_scale(oracle, dtype):
    """16-bit storage on the rare paths: rows peaking ~70 nats apart (direct log-sum-exp in the Z epilogue, far cells
    added through the compare-and-swap atomic), with a per-sample grad_output."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    rng = np.random.default_rng(77)
    N, T, U, A = 2, 6, 4, 50
    f = (rng.standard_normal((N, T, A)) * 0.5).astype(np.float32)
    g = (rng.standard_normal((N, U, A)) * 0.5).astype(np.float32)
    f[:, :, 3] += 70.0
    g[:, :, 11] += 70.0
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl, ll = np.array([T, 4], np.int32), np.array([U - 1, 2], np.int32)
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev).to(dtype).requires_grad_(True)
    tg = torch.tensor(g, device=dev).to(dtype).requires_grad_(True)
    loss = RNNTLossAdd(blank=0, reduction="none")(tf, tg, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev),
                                                  torch.tensor(ll, device=dev))
    w = torch.tensor([0.5, -2.0], device=dev)
    (loss * w).sum().backward()
    fr, gr = tf.detach().double().cpu().numpy(), tg.detach().double().cpu().numpy()
    ref_c, ref_gz = oracle.rnnt_logits(fr[:, :, None, :] + gr[:, None, :, :], labels, tl, ll, 0)
    ref_gz = ref_gz * w.cpu().numpy()[:, None, None, None]
    assert np.abs(loss.detach().double().cpu().numpy() - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10          # a few roundings of the stored value (atomic adds)
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    assert (np.abs(tf.grad.double().cpu().numpy() - rdf) <= 1e-3 + ulp * (np.abs(rdf) + 1.0)).all()
    assert (np.abs(tg.grad.double().cpu().numpy() - rdg) <= 1e-3 + ulp * (np.abs(rdg) + 1.0)).all()


def _bf16_case(oracle, f, g, labels, tl, ll, blank, weights=None, dtype=torch.bfloat16):
    """RNNTLossAdd on bf16 tensors against the oracle on the ROUNDED inputs (loss 1e-4 relative; gradients within the
    storage rounding of the exact sums, the bound of test_sixteen_bit_activations)."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    dev = torch.device("cuda:0")
    tf = torch.tensor(f, device=dev).to(dtype).requires_grad_(True)
    tg = torch.tensor(g, device=dev).to(dtype).requires_grad_(True)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11          # half an ulp of the STORED gradient, relative
    loss = RNNTLossAdd(blank=blank, reduction="none")(tf, tg, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev),
                                                      torch.tensor(ll, device=dev))
    w = torch.ones_like(loss) if weights is None else torch.tensor(weights, device=dev, dtype=loss.dtype)
    (loss * w).sum().backward()
    fr, gr = tf.detach().double().cpu().numpy(), tg.detach().double().cpu().numpy()
    ref_c, ref_gz = oracle.rnnt_logits(fr[:, :, None, :] + gr[:, None, :, :], labels, tl, ll, blank)
    ref_gz = ref_gz * w.double().cpu().numpy()[:, None, None, None]
    assert np.abs(loss.detach().double().cpu().numpy() - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    T, U = f.shape[1], g.shape[1]
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    scale = float(np.abs(w.cpu().numpy()).max())
    edf = np.abs(tf.grad.double().cpu().numpy() - rdf) - (2e-4 * scale * max(1.0, U / 32) + ulp * np.abs(rdf) + 1e-6)
    edg = np.abs(tg.grad.double().cpu().numpy() - rdg) - (2e-4 * scale * max(1.0, T / 32) + ulp * np.abs(rdg) + 1e-6)
    assert edf.max() <= 0, (edf.max(), np.unravel_index(edf.argmax(), edf.shape))
    assert edg.max() <= 0, (edg.max(), np.unravel_index(edg.argmax(), edg.shape))
    return tf.grad, tg.grad


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sixteen_bit_far_cells_accumulate_in_fp32(oracle, dtype):
    """16-bit gradient storage, HUNDREDS of far cells per gradient element (rows peaking 70-90 nats apart on a 520 x 300 lattice;
    tools/add_network_fuzz.py seed 11 case 262): added one cell at a time, every addition rounded to fp16, a dg element came out
    1.7354 for 1.7591.  joint_far16_kernel sums a row segment's far cells in fp32 first: within a few storage quanta now."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    rng = np.random.default_rng(262)
    N, T, U, A, blank = 1, 520, 300, 3, 1
    dev = torch.device("cuda:0")
    f = torch.tensor(rng.standard_normal((N, T, A)) * 1.5, dtype=dtype, device=dev)
    g = torch.tensor(rng.standard_normal((N, U, A)) * 1.5, dtype=dtype, device=dev)
    f[0, ::3, 0] += 70.0
    g[0, ::2, 2] += 90.0
    labels = rng.integers(0, A, size=(N, U - 1))
    labels[labels == blank] = (blank + 1) % A
    tl, ll = np.array([T], np.int32), np.array([U - 1], np.int32)
    z = f.double().cpu().numpy()[:, :, None, :] + g.double().cpu().numpy()[:, None, :, :]
    ref_c, ref_gz = oracle.rnnt_logits(z, labels.astype(np.int32), tl, ll, blank)
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    fa, ga = f.clone().requires_grad_(True), g.clone().requires_grad_(True)
    loss = RNNTLossAdd(blank=blank, reduction="sum")(fa, ga, torch.tensor(labels.astype(np.int32), device=dev),
                                                     torch.tensor(tl, device=dev), torch.tensor(ll, device=dev))
    loss.sum().backward()
    assert abs(loss.item() - ref_c.sum()) <= 1e-4 * abs(ref_c.sum())
    quantum = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7          # relative spacing of the storage type
    for got, ref in ((fa.grad, rdf), (ga.grad, rdg)):
        err = np.abs(got.double().cpu().numpy() - ref)
        assert (err <= 6 * quantum * np.maximum(np.abs(ref), 1.0) + 3e-3 * np.maximum(np.abs(ref), 1.0)).all()


@pytest.mark.parametrize("shape", [
    (2, 33, 21, 512),      # smallest vocabulary of the bf16 matrix-core kernels; second contraction step half masked
    (3, 17, 9, 520),       # a partial last column block (520 = 4 x 128 + 8), T and U below one step
    (2, 50, 41, 1024),     # three steps over the label rows, the last one clamped to the weight row's end (Upad = 48)
    (2, 16, 16, 640),      # exactly one step either way
    (2, 47, 33, 2048),     # one label row past two steps; vocabulary split over wavefronts in the Z kernel
    (1, 150, 21, 5000),    # the benchmark shape of one sample
    (2, 130, 70, 768),     # several tiles of Z in both directions, ragged lengths
    (4, 9, 3, 4096),       # tiny lattice, wide vocabulary
])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bf16_matrix_core_kernels(oracle, shape, dtype):
    """16-bit storage (bf16 or fp16) with rows of whole 16-byte packets and 512 symbols or more runs Z, DF and DG on
    v_mfma_f32_32x32x16_bf16 (rnnt_joint16_kernels.h: operands split into bf16 hi + lo whatever the storage type, fragments
    packed from the packets).  Structural edges of those
    kernels: contraction steps of sixteen rows with masked halves, the clamped second half of a weight row, partial
    column blocks, labels that fall -- or do not fall -- into a wavefront's columns (dense labels on a small range
    included), blank at the first / a middle / the last column, ragged lengths, per-sample grad_output."""
    N, T, U, A = shape
    f, g, labels, tl, ll, blank = problem(shape, sum(shape) + 11)
    _bf16_case(oracle, f, g, labels, tl, ll, blank, dtype=dtype)
    # labels crowded into one column block (every step of DF's label pass hits), blank in the last column, weighted samples
    rng = np.random.default_rng(A + U)
    labels2 = rng.integers(128, 128 + 40, size=labels.shape).astype(np.int32)
    labels2[:, ::3] = labels2[:, :1]                               # repeated labels
    _bf16_case(oracle, f, g, labels2, tl, ll, A - 1, weights=np.linspace(0.5, -1.5, N), dtype=dtype)
    # blank in the first column
    labels3 = labels.copy(); labels3[labels3 == 0] = 1
    _bf16_case(oracle, f, g, labels3, tl, ll, 0, dtype=dtype)


def test_bf16_matrix_core_kernels_on_peaked_rows(oracle):
    """Peaked distributions are where a single bf16 operand would show (few significant terms in Z, cancellation between the
    GEMM term and its correction in the blank / label columns): logits scaled by 6, and rows that exceed their sampled
    reference -- inside the guard (sampled reference kept) and beyond it (exact pass), both through the bf16 kernels."""
    shape = (2, 40, 24, 1024)
    f, g, labels, tl, ll, blank = problem(shape, 5)
    _bf16_case(oracle, f * 4.0, g * 4.0, labels, tl, ll, blank)
    for bump in (20.0, 70.0):
        f2, g2 = f.copy(), g.copy()
        f2[:, ::3, 700] += bump                                     # beyond the first 32 columns: not in the sampled reference
        g2[:, 1::2, 900] += bump
        _bf16_case(oracle, f2, g2, labels, tl, ll, blank)


def test_bf16_matrix_core_kernels_with_masked_vocabulary(oracle):
    """-inf logits through the bf16 matrix-core kernels: masked columns that are neither the blank nor a label (probability
    zero, gradient exactly zero, no NaN), including rows whose first 32 columns -- the sampled reference -- are ALL masked."""
    shape = (2, 37, 19, 640)
    f, g, labels, tl, ll, blank = problem(shape, 21)
    labels = (100 + labels % 50).astype(np.int32)
    blank = 90
    f[..., 300:340] = -np.inf
    g[:, :, 500:560] = -np.inf
    f[:, 3::7, :32] = -np.inf                                  # no finite sample in these rows: the exact pass takes over
    g[:, 2::5, :32] = -np.inf
    df, dg = _bf16_case(oracle, f, g, labels, tl, ll, blank)
    df, dg = df.float().cpu().numpy(), dg.float().cpu().numpy()
    assert not np.isnan(df).any() and not np.isnan(dg).any()
    for sl in (slice(300, 340), slice(500, 560)):
        assert not df[..., sl].any() and not dg[..., sl].any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_size_c3_shape(oracle, dtype):
    """The additive joint at the benchmark's size (N=128, T=150, U=21, A=5000; tools/add_network_bench.py): the first two
    samples against the fp64 oracle on the materialised joint of the (rounded) inputs, ragged lengths on top, and over the
    whole batch the size-independent properties -- every row of df and of dg sums to zero (each row of the joint's gradient
    does), padded rows are exactly zero, costs finite."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    N, T, U, A = 128, 150, 21, 5000
    dev = torch.device("cuda:0")
    g0 = torch.Generator(device=dev).manual_seed(3)
    f = torch.rand((N, T, A), generator=g0, device=dev).to(dtype).requires_grad_(True)
    g = torch.rand((N, U, A), generator=g0, device=dev).to(dtype).requires_grad_(True)
    labels = torch.randint(1, A, (N, U - 1), generator=g0, device=dev, dtype=torch.int32)
    tl = torch.randint(T // 2, T + 1, (N,), generator=g0, device=dev, dtype=torch.int32)
    ll = torch.randint((U - 1) // 2, U, (N,), generator=g0, device=dev, dtype=torch.int32)
    tl[0], ll[0] = T, U - 1
    tl[1], ll[1] = T - 37, 9
    loss = RNNTLossAdd(blank=0, reduction="none")(f, g, labels, tl, ll)
    loss.sum().backward()
    assert bool(torch.isfinite(loss).all())
    df, dg = f.grad.float(), g.grad.float()
    bound = 2e-3 if dtype == torch.float32 else 0.5           # bf16: A roundings of 2^-9 relative each
    assert df.sum(-1).abs().max().item() < bound and dg.sum(-1).abs().max().item() < bound
    t_idx = torch.arange(T, device=dev).view(1, T)
    u_idx = torch.arange(U, device=dev).view(1, U)
    assert df[(t_idx >= tl.view(N, 1))].abs().max().item() == 0.0
    assert dg[(u_idx > ll.view(N, 1))].abs().max().item() == 0.0
    K = 2
    fr, gr = f.detach()[:K].double().cpu().numpy(), g.detach()[:K].double().cpu().numpy()
    z = fr[:, :, None, :] + gr[:, None, :, :]
    oracle.lib().oracle_set_num_threads(K)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels[:K].cpu().numpy(), tl[:K].cpu().numpy(), ll[:K].cpu().numpy(), 0)
    assert np.abs(loss[:K].detach().double().cpu().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    ulp = 0.0 if dtype == torch.float32 else 2.0 ** -8
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    assert (np.abs(df[:K].double().cpu().numpy() - rdf) <= 2e-4 + 5e-5 * np.abs(rdf) + ulp * np.abs(rdf)).all()
    assert (np.abs(dg[:K].double().cpu().numpy() - rdg) <= 2e-4 * (T / 32) + 5e-5 * np.abs(rdg) + ulp * np.abs(rdg)).all()


@pytest.mark.parametrize("loader", ["ext", "ctypes"])
def test_both_bindings(monkeypatch, oracle, loader):
    """RNNTLossAdd through the compiled extension module's C++ autograd function (csrc/binding.cpp: rnnt_loss_add) and through
    the ctypes twin (add_network._RNNTAdd): the three reductions with a per-sample grad_output against the oracle on the
    materialised joint, a gradient for only ONE of the two activations, a no-grad call, FastEmit, 16-bit storage, and the
    argument errors (same exception types and texts from both)."""
    from warprnnt_pytorch import warp_rnnt
    from warprnnt_pytorch.add_network import RNNTLossAdd
    if loader == "ext":
        assert warp_rnnt.binding() == "ext", "the compiled extension module was not built"
    else:
        monkeypatch.setattr(warp_rnnt, "_EXT", None)
    shape = (3, 20, 9, 40)
    f, g, labels, tl, ll, blank = problem(shape, 11)
    N = shape[0]
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    rdf, rdg = ref_gz.sum(axis=2), ref_gz.sum(axis=1)
    dev = torch.device("cuda:0")
    lab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))

    def leaves(fg=True, gg=True, dtype=torch.float32):
        return (torch.tensor(f, device=dev, dtype=dtype).requires_grad_(fg), torch.tensor(g, device=dev, dtype=dtype).requires_grad_(gg))

    tf, tg = leaves()
    per = RNNTLossAdd(blank=blank, reduction="none")(tf, tg, lab, ttl, tll)
    assert per.shape == (N,) and per.dtype == torch.float32
    w = torch.tensor([0.5, -2.0, 3.0], device=dev)
    (per * w).sum().backward()
    wn = w.cpu().numpy().astype(np.float64)
    assert np.allclose(per.detach().cpu().numpy(), ref_c, rtol=1e-5)
    assert np.allclose(tf.grad.cpu().numpy(), rdf * wn[:, None, None], rtol=2e-4, atol=2e-4)
    assert np.allclose(tg.grad.cpu().numpy(), rdg * wn[:, None, None], rtol=2e-4, atol=2e-4)
    for red, val, sc in (("sum", ref_c.sum(), 1.0), ("mean", ref_c.mean(), 1.0 / N)):
        tf, tg = leaves()
        loss = RNNTLossAdd(blank=blank, reduction=red)(tf, tg, lab, ttl, tll)
        assert loss.shape == (1,) and np.allclose(loss.item(), val, rtol=1e-5)
        loss.sum().backward()
        assert np.allclose(tf.grad.cpu().numpy(), rdf * sc, rtol=2e-4, atol=2e-4)
        assert np.allclose(tg.grad.cpu().numpy(), rdg * sc, rtol=2e-4, atol=2e-4)
    tf, tg = leaves(fg=False)                           # only the prediction network trains
    RNNTLossAdd(blank=blank, reduction="sum")(tf, tg, lab, ttl, tll).sum().backward()
    assert tf.grad is None and np.allclose(tg.grad.cpu().numpy(), rdg, rtol=2e-4, atol=2e-4)
    with torch.no_grad():
        tf, tg = leaves()
        assert np.allclose(RNNTLossAdd(blank=blank, reduction="sum")(tf, tg, lab, ttl, tll).item(), ref_c.sum(), rtol=1e-5)
    tf, tg = leaves(dtype=torch.bfloat16)
    loss = RNNTLossAdd(blank=blank, reduction="sum")(tf, tg, lab, ttl, tll)
    loss.sum().backward()
    assert loss.dtype == torch.float32 and tf.grad.dtype == torch.bfloat16 and tg.grad.dtype == torch.bfloat16
    assert abs(loss.item() - ref_c.sum()) < 0.02 * ref_c.sum()
    tf, tg = leaves()
    fe = RNNTLossAdd(blank=blank, reduction="sum", fastemit_lambda=0.5)(tf, tg, lab, ttl, tll)
    fe.sum().backward()
    assert np.allclose(fe.item(), ref_c.sum(), rtol=1e-5)            # FastEmit changes the gradients, not the reported loss
    assert not np.allclose(tg.grad.cpu().numpy(), rdg, rtol=2e-4, atol=2e-4)
    tf, tg = leaves()
    mod = RNNTLossAdd(blank=blank)
    with pytest.raises(ValueError, match="Input length mismatch"):
        mod(tf, tg, lab, ttl - 1, tll)
    with pytest.raises(ValueError, match="Output length mismatch"):
        mod(tf, tg, lab, ttl, tll - 1)
    with pytest.raises(TypeError, match="labels must be"):
        mod(tf, tg, lab.long(), ttl, tll)
    with pytest.raises(ValueError, match="must be contiguous"):
        mod(tf.transpose(1, 2).contiguous().transpose(1, 2), tg, lab, ttl, tll)
    with pytest.raises(ValueError, match="disagree"):
        mod(tf, tg[:, :, :-1].contiguous(), lab, ttl, tll)
    with pytest.raises(TypeError, match="must both be"):
        mod(tf, tg.double(), lab, ttl, tll)
    with pytest.raises(ValueError, match="GPU only"):
        mod(tf.detach().cpu(), tg.detach().cpu(), lab.cpu(), ttl.cpu(), tll.cpu())
