"""-m gpu: NON-FINITE logits.  A NaN or +inf logit (or a row of -inf only) in a row INSIDE a sample's T_b x U_b lattice
makes that sample's cost NaN in the reference -- its max / exp-sum reductions and log_sum_exp propagate
(include/detail/reduce.h:85,103 -> gpu_rnnt_kernel.h:5-9 -> rnnt_helper.h:16-24; the CPU path behind log_softmax does the
same: the oracle returns NaN) -- and NaN gradients on all its in-lattice rows; padded rows stay zero and no other sample
changes.  The HIP path keeps clamped log-probs in its lattice, so it carries the fact separately (rnnt_kernels.h:
note_non_finite / hint_is_poison); these tests pin the behaviour for every statistics-kernel form, both lattice kernels,
the packed layout, the additive joint and the autograd module, and the stale-hint cases of a recycled workspace."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def gpu_call(acts_t, labels, tl, ll, blank=0, want_grad=True, workspace=None):
    """compute_rnnt_loss[_fp64|_bf16] through warp_rnnt.gpu_rnnt; returns (costs float64, grads float64 | None)."""
    from warprnnt_pytorch import warp_rnnt
    lab = torch.tensor(np.asarray(labels, dtype=np.int32), device=DEV)
    t_tl = torch.tensor(np.asarray(tl, dtype=np.int32), device=DEV)
    t_ll = torch.tensor(np.asarray(ll, dtype=np.int32), device=DEV)
    costs = torch.zeros(acts_t.shape[0], dtype=acts_t.dtype if acts_t.dtype == torch.float64 else torch.float32)
    grads = torch.full_like(acts_t, 123.0) if want_grad else torch.zeros(0, device=DEV, dtype=acts_t.dtype)
    assert warp_rnnt.gpu_rnnt(acts_t, lab, t_tl, t_ll, costs, grads, blank, 0, workspace=workspace) == 0   # status: success
    torch.cuda.synchronize()
    return costs.double().numpy(), (grads.double().cpu().numpy() if want_grad else None)


def problem(shape, seed):
    N, T, U, A = shape
    rng = np.random.default_rng(seed)
    acts = rng.standard_normal(shape).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl = np.full(N, T, dtype=np.int32); ll = np.full(N, U - 1, dtype=np.int32)
    if N > 2:                                  # sample 1 -- the one that gets the bad value -- is shorter than the tensor
        tl[1] = max(1, T - 2); ll[1] = max(0, U - 2)
        tl[2] = max(1, T - 1)
    return acts, labels, tl, ll


# (N, T, U, A), dtype: every statistics-kernel form and every lattice form
CASES = [
    ((3, 12, 5, 28), torch.float32),       # LDS-tile statistics, linear-domain lattice (one wavefront, few blocks)
    ((3, 12, 5, 28), torch.bfloat16),
    ((3, 10, 4, 1024), torch.float32),     # 4 KB rows: LDS tile
    ((3, 10, 4, 1024), torch.bfloat16),    # 2 KB rows
    ((3, 8, 4, 1500), torch.float32),      # 6 KB rows: wavefront per row
    ((3, 6, 3, 5000), torch.float32),      # 20 KB rows: block per row
    ((3, 6, 3, 5000), torch.bfloat16),     # 10 KB rows: wavefront per row
    ((3, 20, 70, 50), torch.float32),      # 2-D cell tiles, two-wavefront lattice, tiled coefficient kernel
    ((3, 20, 70, 50), torch.bfloat16),     # (100-byte rows are not whole 8-byte words: the flat tile kernel)
    ((3, 9, 300, 12), torch.float32),      # two lattice columns per lane
    ((3, 7, 4, 33), torch.float64),        # fp64 lattice
    ((300, 5, 3, 9), torch.float32),       # more blocks than compute units: the log-domain one-wavefront lattice
]


@pytest.mark.parametrize("bad", [np.nan, np.inf], ids=["nan", "inf"])
@pytest.mark.parametrize("where", ["blank", "label", "other"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s" % ("x".join(map(str, c[0])), str(c[1]).split(".")[-1]))
def test_non_finite_logits(oracle, case, where, bad):
    shape, dtype = case
    N, T, U, A = shape
    acts, labels, tl, ll = problem(shape, sum(shape))
    b, t, u = 1, int(tl[1]) // 2, min(1, int(ll[1]))          # a cell inside sample 1's lattice
    lab_col = int(labels[b, min(u, U - 2)])
    col = {"blank": 0, "label": lab_col, "other": next(k for k in range(1, A) if k != lab_col)}[where]
    x_clean = torch.tensor(acts, device=DEV).to(dtype)
    x_bad = x_clean.clone(); x_bad[b, t, u, col] = bad
    c0, g0 = gpu_call(x_clean, labels, tl, ll)
    c1, g1 = gpu_call(x_bad, labels, tl, ll)
    assert np.isfinite(c0).all() and np.isfinite(g0).all()
    others = np.arange(N) != b
    assert np.isnan(c1[b])
    assert np.array_equal(c1[others], c0[others]) and np.array_equal(g1[others], g0[others])   # bit for bit
    Tb, Ub = int(tl[b]), int(ll[b]) + 1
    assert np.isnan(g1[b, :Tb, :Ub]).all()                                    # every in-lattice row of the sample
    assert not g1[b, Tb:].any() and not g1[b, :, Ub:].any()                   # padding: zeros
    # the oracle (= the reference's arithmetic) agrees on what is NaN
    ref_c, ref_g = oracle.rnnt_logits(x_bad.double().cpu().numpy(), labels, tl, ll)
    assert np.array_equal(np.isnan(ref_c), np.isnan(c1)) and np.array_equal(np.isnan(ref_g), np.isnan(g1))
    # score only (alpha sweep alone)
    c2, _ = gpu_call(x_bad, labels, tl, ll, want_grad=False)
    assert np.isnan(c2[b]) and np.array_equal(c2[others], c0[others])
    # the same value in a PADDED row changes nothing
    if Tb < T:
        x_pad = x_clean.clone(); x_pad[b, T - 1, 0, col] = bad
        c3, g3 = gpu_call(x_pad, labels, tl, ll)
        assert np.array_equal(c3, c0) and np.array_equal(g3, g0)


@pytest.mark.parametrize("case", [CASES[0], CASES[4], CASES[7]], ids=["tile", "wave", "tile2d"])
def test_row_of_minus_inf_only(oracle, case):
    """All logits of a row -inf: no distribution exists; the reference's exp(x - max) is exp(nan)."""
    shape, dtype = case
    acts, labels, tl, ll = problem(shape, 3)
    x = torch.tensor(acts, device=DEV).to(dtype)
    x[1, 1, 0, :] = -np.inf
    c, g = gpu_call(x, labels, tl, ll)
    ref_c, ref_g = oracle.rnnt_logits(x.double().cpu().numpy(), labels, tl, ll)
    assert np.isnan(c[1]) and np.array_equal(np.isnan(ref_c), np.isnan(c)) and np.array_equal(np.isnan(ref_g), np.isnan(g))
    keep = np.arange(shape[0]) != 1
    assert np.abs(c[keep] - ref_c[keep]).max() < 1e-4 * np.abs(ref_c[keep]).max()


@pytest.mark.parametrize("case", [CASES[0], CASES[7], CASES[11]], ids=["linear", "wide", "log-domain"])
def test_stale_hints_in_a_recycled_workspace(case):
    """The workspace is undefined on entry: hint words full of garbage, or left over from a poisoned call of another
    shape, must never poison a clean sample -- and a poisoned call must still be seen on such a workspace."""
    from warprnnt_pytorch import _lib
    shape, dtype = case
    N, T, U, A = shape
    acts, labels, tl, ll = problem(shape, 11)
    x = torch.tensor(acts, device=DEV).to(dtype)
    c0, g0 = gpu_call(x, labels, tl, ll)
    nbytes = _lib.workspace_bytes(T, U, N, True, 4)
    gen = torch.Generator(device=DEV); gen.manual_seed(5)
    for fill in ("random", "ones", "small", "nan"):
        if fill == "random":
            ws = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=DEV, generator=gen)
        elif fill == "ones":
            ws = torch.full((nbytes,), 255, dtype=torch.uint8, device=DEV)
        elif fill == "small":      # every word a plausible hint: small positive integers
            ws = torch.randint(1, 4000, (nbytes // 4,), dtype=torch.int32, device=DEV, generator=gen).view(torch.uint8)
        else:                      # every float a NaN: a hinted cell that this call does not write reads as non-finite
            ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=DEV).view(torch.uint8)
        if ws.numel() < nbytes:
            ws = torch.cat([ws, torch.zeros(nbytes - ws.numel(), dtype=torch.uint8, device=DEV)])
        c, g = gpu_call(x, labels, tl, ll, workspace=ws)
        assert np.array_equal(c, c0) and np.array_equal(g, g0), fill
        xb = x.clone(); xb[0, 0, 0, 2] = np.nan
        c, _ = gpu_call(xb, labels, tl, ll, workspace=ws)
        assert np.isnan(c[0]) and np.array_equal(c[1:], c0[1:]), fill
        c, g = gpu_call(x, labels, tl, ll, workspace=ws)              # and clean again right behind it
        assert np.array_equal(c, c0) and np.array_equal(g, g0), fill


def test_module_and_packed_layout():
    from warprnnt_pytorch import RNNTLoss
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
    shape = (4, 11, 6, 37)
    acts, labels, tl, ll = problem(shape, 8)
    t_lab, t_tl, t_ll = (torch.tensor(v, device=DEV) for v in (labels, tl, ll))
    for bad in (float("nan"), float("inf")):
        res = {}
        for poisoned in (False, True):
            a = torch.tensor(acts, device=DEV)
            if poisoned:
                a[1, 2, 1, 5] = bad
            a.requires_grad_(True)
            loss = RNNTLoss(reduction="none")(a, t_lab, t_tl, t_ll)
            loss.sum().backward()
            p = pack_joint(a.detach(), t_tl, t_ll).contiguous().requires_grad_(True)
            lp = RNNTLossPacked(reduction="none")(p, t_lab, t_tl, t_ll)
            lp.sum().backward()
            res[poisoned] = (loss.detach().cpu().numpy(), a.grad.cpu().numpy(), lp.detach().cpu().numpy(),
                             pack_joint(a.grad, t_tl, t_ll).cpu().numpy(), p.grad.cpu().numpy())
        (c0, g0, pc0, _, pg0), (c1, g1, pc1, gp1, pg1) = res[False], res[True]
        keep = np.arange(shape[0]) != 1
        for c_clean, c_bad in ((c0, c1), (pc0, pc1)):
            assert np.isnan(c_bad[1]) and np.array_equal(c_bad[keep], c_clean[keep])
        assert np.array_equal(g1[keep], g0[keep]) and np.isnan(g1[1, :tl[1], :ll[1] + 1]).all()
        assert np.array_equal(np.isnan(pg1), np.isnan(gp1))                     # packed gradient: NaN on the same rows
        assert np.array_equal(pg1[~np.isnan(pg1)], pg0[~np.isnan(pg1)])
        # 'mean' / 'sum' carry the NaN into the scalar loss, which is what a training loop's isfinite() guard reads
        a = torch.tensor(acts, device=DEV); a[1, 2, 1, 5] = bad
        assert torch.isnan(RNNTLoss()(a, t_lab, t_tl, t_ll)).all()


@pytest.mark.parametrize("shape", [(3, 20, 9, 40), (3, 33, 21, 257), (3, 40, 16, 64), (3, 40, 35, 2048), (3, 9, 70, 7)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("bad", [np.nan, np.inf], ids=["nan", "inf"])
def test_additive_joint(shape, bad):
    """compute_rnnt_loss_add: a bad value in f[b,t,:] (or g[b,u,:]) is in every cell of that time (label) row."""
    from warprnnt_pytorch.add_network import RNNTLossAdd
    N, T, U, A = shape
    rng = np.random.default_rng(sum(shape))
    f = rng.standard_normal((N, T, A)).astype(np.float32)
    g = rng.standard_normal((N, U, A)).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl = np.array([T, T - 2, T - 1], dtype=np.int32); ll = np.array([U - 1, U - 2, U - 1], dtype=np.int32)

    def run(f_np, g_np, dtype=torch.float32):
        tf = torch.tensor(f_np, device=DEV).to(dtype).requires_grad_(True)
        tg = torch.tensor(g_np, device=DEV).to(dtype).requires_grad_(True)
        loss = RNNTLossAdd(reduction="none")(tf, tg, torch.tensor(labels, device=DEV), torch.tensor(tl, device=DEV),
                                             torch.tensor(ll, device=DEV))
        loss.sum().backward()
        return loss.detach().float().cpu().numpy(), tf.grad.float().cpu().numpy(), tg.grad.float().cpu().numpy()

    for dtype in (torch.float32, torch.bfloat16):
        c0, df0, dg0 = run(f, g, dtype)
        assert np.isfinite(c0).all()
        for which in ("f", "g"):
            fb, gb = f.copy(), g.copy()
            if which == "f":
                fb[1, 3, 4] = bad
            else:
                gb[1, 2, 4] = bad
            c1, df1, dg1 = run(fb, gb, dtype)
            keep = np.arange(N) != 1
            assert np.isnan(c1[1]), (dtype, which)
            # (not bit for bit here: an +inf trips the guard of the sampled row references, and the exact pass that then
            # re-runs covers the whole batch -- the other samples come out of a differently rounded, equally valid pass)
            rt = 1e-4 if dtype == torch.float32 else 1.6e-2           # (bf16 gradients: two storage quanta)
            assert np.allclose(c1[keep], c0[keep], rtol=1e-6) and np.allclose(df1[keep], df0[keep], rtol=rt, atol=rt * 0.1) \
                and np.allclose(dg1[keep], dg0[keep], rtol=rt, atol=rt * 0.1)
            assert np.isfinite(df1[keep]).all() and np.isfinite(dg1[keep]).all()
            assert np.isnan(df1[1, :tl[1]]).any() and np.isnan(dg1[1, :ll[1] + 1]).any()


# ---------------------------------------------------------------------------------------------------------------------------
# No alignment with non-zero probability: a REQUIRED label masked with -inf wherever it could be emitted.  The reference's
# arithmetic ends on ll = -inf: cost +inf, every in-lattice gradient exp(-inf + inf) = NaN (rnnt_helper.h:16-24,
# gpu_rnnt_kernel.h:161-174; the oracle restates exactly that).  The lattice kernels work with a finite "log zero" sentinel, so
# the sweep ends on ~ -1e30 instead -- which used to come out as a finite cost of 6.9e29 with finite gradients.
@pytest.mark.parametrize("case", [(3, 6, 5, 8, torch.float32), (2, 40, 70, 20, torch.float32), (3, 9, 4, 1024, torch.float32),
                                  (2, 30, 300, 12, torch.float32), (3, 6, 5, 8, torch.float64), (3, 12, 7, 64, torch.bfloat16)],
                         ids=["one-wavefront", "three-wavefront", "long-rows", "two-column-lanes", "fp64", "bf16"])
def test_impossible_alignment_is_an_infinite_cost(oracle, case):
    N, T, U, A, dtype = case
    acts, labels, tl, ll = problem((N, T, U, A), T * U + A)
    victim = 1
    u = int(ll[victim]) // 2                                # a label the victim must emit (u < its label length)
    assert u < int(ll[victim])
    x_clean = torch.tensor(acts, device=DEV).to(dtype)
    x_bad = x_clean.clone()
    x_bad[victim, :, u, int(labels[victim, u])] = -float("inf")   # ... and cannot, at any time step
    ref_c, ref_g = oracle.rnnt_logits(x_bad.double().cpu().numpy(), labels, tl, ll)
    Tb, Ub = int(tl[victim]), int(ll[victim]) + 1
    assert np.isposinf(ref_c[victim]) and np.isnan(ref_g[victim, :Tb, :Ub]).all()
    c0, g0 = gpu_call(x_clean, labels, tl, ll)
    c1, g1 = gpu_call(x_bad, labels, tl, ll)
    assert np.isposinf(c1[victim]), c1
    assert np.isnan(g1[victim, :Tb, :Ub]).all()
    assert not g1[victim, Tb:].any() and not g1[victim, :, Ub:].any()          # padding: zeros
    others = np.arange(N) != victim
    assert np.array_equal(c1[others], c0[others]) and np.array_equal(g1[others], g0[others])
    assert np.isfinite(c1[others]).all()
    c2, _ = gpu_call(x_bad, labels, tl, ll, want_grad=False)                   # score only
    assert np.isposinf(c2[victim]) and np.array_equal(c2[others], c0[others])


def test_impossible_alignment_additive_joint_and_packed(oracle):
    from warprnnt_pytorch.add_network import RNNTLossAdd
    from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
    rng = np.random.default_rng(5)
    N, T, U, A, blank = 2, 12, 70, 20, 0
    dev = torch.device("cuda:0")
    f = rng.standard_normal((N, T, A)).astype(np.float32)
    g = rng.standard_normal((N, U, A)).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl, ll = np.array([T, T - 3], np.int32), np.array([U - 1, U - 5], np.int32)
    g[0, 30, labels[0, 30]] = -np.inf                       # sample 0 can never emit its label 30
    lab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    tf, tg = torch.tensor(f, device=dev, requires_grad=True), torch.tensor(g, device=dev, requires_grad=True)
    loss = RNNTLossAdd(blank=blank, reduction="none")(tf, tg, lab, ttl, tll)
    loss.sum().backward()
    c = loss.detach().cpu().numpy()
    assert np.isposinf(c[0]) and np.isfinite(c[1])
    assert torch.isnan(tf.grad[0]).all() and torch.isfinite(tf.grad[1]).all() and torch.isfinite(tg.grad[1]).all()
    z = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    ref_c, _ = oracle.rnnt_logits(z, labels, tl, ll, blank)
    assert np.isposinf(ref_c[0]) and abs(c[1] - ref_c[1]) <= 1e-4 * abs(ref_c[1])
    joint = torch.tensor(z.astype(np.float32), device=dev)
    xp = pack_joint(joint, ttl, tll).contiguous().requires_grad_(True)
    lp = RNNTLossPacked(blank=blank, reduction="none")(xp, lab, ttl, tll)
    lp.sum().backward()
    cp = lp.detach().cpu().numpy()
    assert np.isposinf(cp[0]) and abs(cp[1] - ref_c[1]) <= 1e-4 * abs(ref_c[1])
    n0 = int(tl[0]) * (int(ll[0]) + 1)
    assert torch.isnan(xp.grad[:n0]).all() and torch.isfinite(xp.grad[n0:]).all()
