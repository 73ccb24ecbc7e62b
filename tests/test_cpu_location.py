"""RNNT_CPU location of libwarprnnt.so through the C-ABI -- the reference's tests/test_cpu.cpp
restated (small_test :12-71, options_test :73-179, inf_test :181-240, grad_check :287-379) plus
the cases the reference never tests (variable lengths, blank != 0, fp64, !batch_first)."""
import ctypes as C

import numpy as np
import pytest

from tests.golden import literals as G
from tests.golden.make_golden import CASES, case_inputs
from warprnnt_pytorch import _lib

FIX = np.load(__file__.replace("test_cpu_location.py", "golden/ref_cases.npz"))


def cpu_loss(log_probs, labels, act_lens, label_lens, blank=0, want_grad=True, num_threads=1,
             batch_first=True, dims=None):
    lib = _lib.lib()
    lp = np.ascontiguousarray(log_probs)
    N, T, U, A = dims if dims else lp.shape
    fn, esz = (lib.compute_rnnt_loss, 4) if lp.dtype == np.float32 else (lib.compute_rnnt_loss_fp64, 8)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    tl = np.ascontiguousarray(act_lens, dtype=np.int32)
    ll = np.ascontiguousarray(label_lens, dtype=np.int32)
    ws = np.empty(_lib.workspace_bytes(T, U, N, False, esz), dtype=np.uint8)
    costs = np.zeros(N, dtype=lp.dtype)
    grads = np.full_like(lp, 7.0) if want_grad else None
    opt = _lib.rnntOptions(loc=_lib.RNNT_CPU, num_threads=num_threads, stream=None, blank_label=blank,
                           maxT=T, maxU=U, batch_first=batch_first)
    st = fn(lp.ctypes.data, grads.ctypes.data if want_grad else None, labels.ctypes.data, ll.ctypes.data,
            tl.ctypes.data, A, N, costs.ctypes.data, ws.ctypes.data, opt)
    assert st == 0, _lib.status_string(st)
    return costs, grads


def test_small_test(oracle):
    lp = oracle.log_softmax(G.SMALL_ACTS.astype(np.float32))
    c, _ = cpu_loss(lp, G.SMALL_LABELS, [2], [2], want_grad=False)
    assert G.SMALL_COST - 1e-4 < c[0] < G.SMALL_COST + 1e-4


def test_options_test(oracle):
    lp = oracle.log_softmax(G.OPTIONS_ACTS_6DP.astype(np.float32))
    c, g = cpu_loss(lp, G.OPTIONS_LABELS, [4, 4], [2, 2])
    assert np.abs(g - G.OPTIONS_LOGPROB_GRADS).max() < 1e-4
    assert np.abs(c - G.OPTIONS_COSTS).max() < 1e-4


def test_inf_test(oracle):
    # un-normalised acts fed straight in (tests/test_cpu.cpp:219): finite cost, no NaN grads
    acts, labels, tl, ll, blank = case_inputs("inf_test")
    c, g = cpu_loss(acts.astype(np.float32), labels, tl, ll)
    assert np.isfinite(c).all() and not np.isnan(g).any()


@pytest.mark.parametrize("A,T,L,B,tol", [(20, 50, 15, 1, 1e-4), (5, 10, 5, 65, 1e-4)])
def test_grad_check(oracle, A, T, L, B, tol):
    """Central differences, eps 1e-2, rel_diff = sum (g-ng)^2 / sum g^2 < tol (tests/test.h:22-32,
    tests/test_cpu.cpp:242-285,347-351).  Differences are taken on a random subset of elements
    (the reference perturbs every element) to keep the CPU suite fast."""
    acts = oracle.gen_acts(A * T * L * B).reshape(B, T, L, A)            # un-normalised, as the reference
    labels = np.tile(oracle.gen_labels(A, L - 1), (B, 1))
    tl, ll = np.full(B, T), np.full(B, L - 1)
    _, g = cpu_loss(acts, labels, tl, ll)
    rng = np.random.default_rng(0)
    flat = acts.reshape(-1)
    picks = rng.choice(flat.size, size=min(400, flat.size), replace=False)
    num = np.zeros(picks.size)
    for k, i in enumerate(picks):
        old = flat[i]
        flat[i] = old + 1e-2
        cp, _ = cpu_loss(acts, labels, tl, ll, want_grad=False)
        flat[i] = old - 1e-2
        cm, _ = cpu_loss(acts, labels, tl, ll, want_grad=False)
        flat[i] = old
        num[k] = (cp.sum(dtype=np.float64) - cm.sum(dtype=np.float64)) / 2e-2
    an = g.reshape(-1)[picks].astype(np.float64)
    assert ((an - num) ** 2).sum() / (an ** 2).sum() < 5e-3   # fp32 differences on 400 elements


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_against_reference_fixture(oracle, name, dt):
    acts, labels, tl, ll, blank = case_inputs(name)
    lp = oracle.log_softmax(acts.astype(dt))
    c, g = cpu_loss(lp, labels, tl, ll, blank, num_threads=0)
    tol = 1e-10 if dt == np.float64 else 2e-6
    assert np.abs(c - FIX[name + "/costs64"]).max() <= max(tol, 1e-9) * max(1, np.abs(c).max()) + (0 if dt == np.float64 else 2e-4)
    # fp32 location vs the reference fp64 outputs: the reference tests allow 1e-4 (test_cpu.cpp:149)
    assert np.abs(g - FIX[name + "/lpgrad64"]).max() < (1e-6 if dt == np.float64 else 1e-4)
    cf, _ = cpu_loss(lp, labels, tl, ll, blank, want_grad=False)
    assert np.array_equal(cf, c)


def test_time_major_layout(oracle):
    """batch_first = false: (T,U,B,V) layout of the reference (cpu_rnnt.h:140-144,294-295);
    gradients are not zeroed in this mode (only the sparse entries are written)."""
    acts, labels, tl, ll, blank = case_inputs("blank5_a19")
    lp = oracle.log_softmax(acts)
    c_ref, g_ref = cpu_loss(lp, labels, tl, ll, blank)
    N, T, U, A = lp.shape
    lp_t = np.ascontiguousarray(lp.transpose(1, 2, 0, 3))
    c, g = cpu_loss(lp_t, labels, tl, ll, blank, batch_first=False, dims=(N, T, U, A))
    assert np.allclose(c, c_ref)
    g = g.transpose(2, 0, 1, 3)
    touched = g != 7.0
    assert np.allclose(g[touched], g_ref[touched]) and not g_ref[~touched].any()


def test_bad_lengths_are_rejected(oracle):
    lib = _lib.lib()
    lp = np.zeros((1, 2, 3, 4), dtype=np.float32)
    ws = np.empty(4096, dtype=np.uint8)
    costs = np.zeros(1, dtype=np.float32)
    lab = np.ones((1, 2), dtype=np.int32)
    opt = _lib.rnntOptions(loc=0, num_threads=1, stream=None, blank_label=0, maxT=2, maxU=3, batch_first=True)
    def call(tlen, llen):
        tl, ll = np.array([tlen], dtype=np.int32), np.array([llen], dtype=np.int32)   # keep alive
        return lib.compute_rnnt_loss(lp.ctypes.data, None, lab.ctypes.data, ll.ctypes.data, tl.ctypes.data, 4, 1,
                                     costs.ctypes.data, ws.ctypes.data, opt)

    for tlen, llen in ((3, 2), (2, 3), (0, 2)):
        assert call(tlen, llen) == 2
    assert call(2, 2) == 0
    opt.blank_label = 4
    assert call(2, 2) == 2


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_packed_layout_on_the_cpu_location(oracle, dtype):
    """compute_rnnt_loss_packed with RNNT_CPU: ragged samples stored back to back (row = offsets[b] + t*U_b + u),
    every array on the host.  Costs and the sparse log-prob gradients must equal the padded CPU call's on the
    rows that exist; wrong row counts are refused."""
    rng = np.random.default_rng(4)
    N, T, U, A, blank = 4, 9, 5, 6, 2
    lp = oracle.log_softmax(rng.standard_normal((N, T, U, A))).astype(dtype)
    labels = rng.integers(0, A, size=(N, U - 1)).astype(np.int32)
    labels[labels == blank] = (blank + 1) % A
    tl = np.array([9, 3, 7, 1], dtype=np.int32)
    ll = np.array([4, 0, 2, 3], dtype=np.int32)
    c_pad, g_pad = cpu_loss(lp, labels, tl, ll, blank=blank)
    rows = [lp[b, :tl[b], :ll[b] + 1].reshape(-1, A) for b in range(N)]
    packed = np.ascontiguousarray(np.concatenate(rows))
    offs = np.concatenate([[0], np.cumsum([r.shape[0] for r in rows])]).astype(np.int64)
    lib = _lib.lib()
    esz = packed.dtype.itemsize
    ws = np.empty(_lib.workspace_bytes(T, U, N, False, esz), dtype=np.uint8)
    costs = np.zeros(N, dtype=dtype)
    grads = np.full_like(packed, 7.0)
    opt = _lib.rnntOptions(loc=_lib.RNNT_CPU, num_threads=2, stream=None, blank_label=blank, maxT=T, maxU=U,
                           batch_first=True)

    def call(rows_total=int(offs[-1]), offsets=offs, code=0 if dtype == np.float32 else 1, lam=0.0):
        return lib.compute_rnnt_loss_packed(packed.ctypes.data, grads.ctypes.data, labels.ctypes.data, ll.ctypes.data,
                                            tl.ctypes.data, offsets.ctypes.data, rows_total, A, N, costs.ctypes.data,
                                            None, ws.ctypes.data, opt, code, lam)

    assert call() == 0
    assert np.allclose(costs, c_pad, rtol=1e-6)
    expect = np.concatenate([g_pad[b, :tl[b], :ll[b] + 1].reshape(-1, A) for b in range(N)])
    assert np.allclose(grads, expect, rtol=1e-6, atol=1e-7)
    bad = offs.copy(); bad[2] += 1
    assert call(offsets=bad) == 2                     # a sample whose row count is not T_b * U_b
    assert call(rows_total=0) == 2 and call(code=2) == 2 and call(lam=0.1) == 2



@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf], ids=["nan", "inf", "-inf"])
def test_non_finite_log_probs_propagate_as_in_the_reference(oracle, bad):
    """RNNT_CPU takes log-probs as they come (the reference's CPU contract): a NaN or +inf at a transition the lattice uses
    makes that sample's cost NaN through log_sum_exp (include/detail/rnnt_helper.h:16-24), a -inf is a closed transition;
    the other samples are untouched.  Same answer as the reference's own library where it has been built (oracle/_ref)."""
    rng = np.random.default_rng(4)
    N, T, U, A = 3, 6, 4, 7
    lp = oracle.log_softmax(rng.standard_normal((N, T, U, A)))
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl, ll = [T, T - 1, T], [U - 1, U - 2, U - 1]
    c0, g0 = cpu_loss(lp, labels, tl, ll)
    x = lp.copy()
    x[1, 2, 1, 0] = bad                                   # the blank transition out of cell (2, 1) of sample 1
    c, g = cpu_loss(x, labels, tl, ll)
    assert np.array_equal(c[[0, 2]], c0[[0, 2]]) and np.array_equal(g[[0, 2]], g0[[0, 2]])
    if bad == -np.inf:
        assert np.isfinite(c[1]) and c[1] >= c0[1]        # a path less: the likelihood can only drop
    else:
        assert np.isnan(c[1])
    rc, _ = oracle.rnnt_logprobs(x, labels, tl, ll)
    assert np.array_equal(np.isnan(rc), np.isnan(c))
    if oracle.have_ref():
        fc, _ = oracle.ref_rnnt_logprobs(x, labels, tl, ll)
        assert np.array_equal(np.isnan(fc), np.isnan(c))
        assert np.allclose(fc[~np.isnan(fc)], c[~np.isnan(c)], rtol=1e-10)


def test_impossible_alignment_is_an_infinite_cost_on_the_cpu_too(oracle):
    """A label the sample must emit, closed (-inf) at every time step: no alignment is left.  The reference's arithmetic ends on
    ll = -inf -> cost +inf (the GPU location matches it since round 4: tests/test_gpu_non_finite.py); RNNT_CPU, the oracle and
    the reference's own library agree, the other samples are untouched."""
    rng = np.random.default_rng(8)
    N, T, U, A = 3, 6, 4, 7
    lp = oracle.log_softmax(rng.standard_normal((N, T, U, A)))
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl, ll = [T, T - 1, T], [U - 1, U - 2, U - 1]
    c0, g0 = cpu_loss(lp, labels, tl, ll)
    x = lp.copy()
    x[1, :, 0, labels[1, 0]] = -np.inf                    # sample 1 can never emit its first label
    c, g = cpu_loss(x, labels, tl, ll)
    assert np.isposinf(c[1])
    assert np.array_equal(c[[0, 2]], c0[[0, 2]]) and np.array_equal(g[[0, 2]], g0[[0, 2]])
    rc, _ = oracle.rnnt_logprobs(x, labels, tl, ll)
    assert np.isposinf(rc[1]) and np.allclose(rc[[0, 2]], c[[0, 2]], rtol=1e-6)
    if oracle.have_ref():
        fc, _ = oracle.ref_rnnt_logprobs(x, labels, tl, ll)
        assert np.isposinf(fc[1])


def test_label_equal_to_blank_keeps_the_cpu_references_assignment(oracle):
    """A label that equals the blank: the reference's CPU location ASSIGNS the label term after the blank term
    (cpu_rnnt.h:253-267: the second write overwrites the first), its GPU location subtracts both
    (gpu_rnnt_kernel.h:161-174).  RNNT_CPU here keeps the CPU contract, assignment and all (include/rnnt.h says so);
    the GPU location is pinned by tests/test_gpu_label_equals_blank.py against fp64 autograd."""
    import torch
    from warprnnt_pytorch import warp_rnnt
    shape = (2, 9, 5, 7)
    rng = np.random.default_rng(2)
    blank = 3
    labels = rng.integers(0, 7, size=(2, 4)).astype(np.int32)
    labels[:, ::2] = blank
    tl, ll = np.array([9, 6], np.int32), np.array([4, 3], np.int32)
    lp = torch.log_softmax(torch.tensor(rng.standard_normal(shape), dtype=torch.float32), -1)
    costs, grads = torch.zeros(2), torch.zeros(shape)
    assert warp_rnnt.cpu_rnnt(lp, torch.tensor(labels), torch.tensor(tl), torch.tensor(ll), costs, grads, blank, 0) == 0
    ref_c, ref_g = oracle.rnnt_logprobs(lp.numpy(), labels, tl, ll, blank, True)
    assert np.allclose(costs.numpy(), ref_c, rtol=1e-5) and np.abs(grads.numpy() - ref_g).max() < 1e-5
    if oracle.have_ref():
        rc, rg = oracle.ref_rnnt_logprobs(lp.numpy(), labels, tl, ll, blank, True, 1)
        assert np.allclose(costs.numpy(), rc, rtol=1e-5) and np.abs(grads.numpy() - rg).max() < 1e-5
