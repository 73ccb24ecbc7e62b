"""-m gpu: parity at BASELINE.json's full sizes on the REFERENCE's input streams, past 2^31 elements, at the
lattice kernels' structural limits, and the validation the reference leaves undefined.

Tolerances (BASELINE.json north_star; SURVEY.md 8c "tolerance floor"): loss within 1e-4 RELATIVE of the fp64
oracle, gradients within 1e-3 absolute for fp32 storage; bf16 storage 4e-3 = half a bf16 ulp at |g| ~ 1 (2^-8),
the quantum of the STORAGE type -- the arithmetic is the same fp32 as for fp32 storage.  Those absolute figures exceed
every non-blank / non-label entry at A = 5000 / 1024, so each such comparison is followed by the per-element one
(oracle.grad_bound: one rounding of the stored value + the fp32 arithmetic ahead of it, relative to the terms of the
element) and by a negative control where it matters (config 5)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.test_gpu_parity import run_gpu

pytestmark = pytest.mark.gpu


def _ref_stream_batch(oracle, n_ref, N, T, U, A, dev):
    """A (N,T,U,A) fp32 batch whose first `n_ref` samples are the head of the reference harness's mt19937(0)
    stream (tests/random.cpp:4-12: batch is the slowest dimension, so these ARE its samples 0..n_ref-1); the rest
    is device-generated uniform(0,1).  Labels: genLabels(A, L) for every sample (tests/test_time.cu:52-58)."""
    x = torch.empty((N, T, U, A), dtype=torch.float32, device=dev)
    head = oracle.gen_acts(n_ref * T * U * A).reshape(n_ref, T, U, A)
    x[:n_ref] = torch.from_numpy(head).to(dev)
    if n_ref < N:
        g = torch.Generator(device=dev).manual_seed(99)
        x[n_ref:] = torch.rand((N - n_ref, T, U, A), generator=g, device=dev)
    lab = oracle.gen_labels(A, U - 1)
    labels = np.tile(lab, (N, 1)).astype(np.int32)
    return x, head, labels


def _gpu_call(x, labels, tl, ll, ws=None):
    from warprnnt_pytorch import warp_rnnt
    dev = x.device
    costs = torch.zeros(x.shape[0], dtype=torch.float64 if x.dtype == torch.float64 else torch.float32)
    grads = torch.full_like(x, 7.0)
    assert warp_rnnt.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev),
                              torch.tensor(ll, device=dev), costs, grads, 0, 0, workspace=ws) == 0
    return costs, grads


def _likelihoods(ws, N, T, U, dtype=torch.float32):
    """Forward / backward log-likelihoods left in the workspace by the last gradient-computing call
    (compute_rnnt_loss_likelihoods: the reference's llForward / llBackward, include/detail/gpu_rnnt.h:92-105)."""
    from warprnnt_pytorch import _lib
    llf, llb = np.zeros(N), np.zeros(N)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, maxT=T, maxU=U, batch_first=True)
    st = _lib.lib().compute_rnnt_loss_likelihoods(ws.data_ptr(), N, opt, _lib.DT_F64 if dtype == torch.float64 else _lib.DT_F32,
                                                  llf.ctypes.data, llb.ctypes.data)
    assert st == 0, st
    return llf, llb


def test_c2_whole_batch_on_the_reference_stream(oracle):
    """BASELINE config 2 (README row N=16,T=150,L=40,A=28) exactly as tests/test_time.cu feeds it: every element
    of the batch from the reference's generators, every sample against the fp64 oracle."""
    N, T, U, A = 16, 150, 41, 28
    dev = torch.device("cuda:0")
    x, head, labels = _ref_stream_batch(oracle, N, N, T, U, A, dev)
    tl, ll = np.full(N, T, np.int32), np.full(N, U - 1, np.int32)
    costs, grads = _gpu_call(x, labels, tl, ll)
    ref_c, ref_g = oracle.rnnt_logits(head.astype(np.float64), labels, tl, ll)
    assert np.abs(costs.double().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    assert np.abs(grads.double().cpu().numpy() - ref_g).max() < 1e-4
    # the reference library itself (fp32 CPU path + log_softmax + chain rule) where it is available
    if oracle.have_ref():
        lp = oracle.log_softmax(head.astype(np.float64)).astype(np.float32)
        rc, rg = oracle.ref_rnnt_logprobs(lp, labels, tl, ll, 0, True, 4)
        assert np.abs(costs.numpy() - rc).max() <= 1e-4 * np.abs(rc).max()
        rgl = oracle.chain_rule_to_logits(lp.astype(np.float64), rg.astype(np.float64))
        assert np.abs(grads.double().cpu().numpy() - rgl).max() < 1e-3


@pytest.mark.parametrize("name,shape", [("c3", (128, 150, 21, 5000)), ("c4", (64, 1500, 301, 50))])
def test_full_size_samples_on_the_reference_stream(oracle, name, shape):
    """c3 / c4 at full batch size: the first 32 samples are the reference harness's own stream and are checked
    against the fp64 oracle, with variable lengths on top (T_b, U_b of the checked samples span the range)."""
    N, T, U, A = shape
    dev = torch.device("cuda:0")
    K = 32
    x, head, labels = _ref_stream_batch(oracle, K, N, T, U, A, dev)
    rng = np.random.default_rng(4)
    tl = rng.integers(T // 2, T + 1, size=N).astype(np.int32)
    ll = rng.integers((U - 1) // 2, U, size=N).astype(np.int32)
    tl[0], ll[0] = T, U - 1
    tl[1], ll[1] = T, (U - 1) // 2
    tl[2], ll[2] = T // 2, U - 1
    from warprnnt_pytorch import _lib
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    costs, grads = _gpu_call(x, labels, tl, ll, ws)
    # the forward and the backward recursion agree on log P(y|x) for EVERY sample of the batch (the reference's CPU path
    # warns beyond 1e-1 absolute, include/detail/cpu_rnnt.h:167-170; c4 is 1800 diagonals of scaled fp32)
    llf, llb = _likelihoods(ws, N, T, U)
    assert np.abs(llf + costs.double().numpy()).max() <= 2e-7 * np.abs(llf).max()
    assert np.abs(llf - llb).max() <= 1e-5 * np.abs(llf).max(), np.abs(llf - llb).max()
    oracle.lib().oracle_set_num_threads(min(64, os.cpu_count() or 8))
    ref_c, ref_g, mag = oracle.rnnt_logits(head.astype(np.float64), labels[:K], tl[:K], ll[:K], want_mag=True)
    got_c, got_g = costs[:K].double().numpy(), grads[:K].double().cpu().numpy()
    assert np.abs(got_c - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    assert np.abs(got_g - ref_g).max() < 1e-3                          # north_star's absolute figure ...
    # ... which is larger than every non-blank / non-label entry at A = 5000 (<= 3.2e-4): per element, relative
    oracle.assert_grads(got_g, ref_g, mag, torch.float32, what=name)
    assert torch.isfinite(costs).all()
    assert grads.sum(-1).abs().max().item() < 2e-4                     # every row of the logit gradient sums to 0
    assert (grads.sum(-1).abs() / oracle.rowsum_bound(grads.abs().sum(-1), torch.float32)).max().item() <= 1.0


def test_more_than_2_31_elements_on_one_gpu(oracle):
    """BASELINE config 5 UNSHARDED: N=1024,T=200,U=41,A=1024 bf16 = 8.6e9 elements (the reference indexes with
    32-bit int: include/detail/gpu_rnnt_kernel.h:7-8,161,174).  17.2 GB of logits + 17.2 GB of gradients on one
    288 GB MI355X; oracle on samples {0, 511, 1023}; row sums and padding over the whole tensor."""
    N, T, U, A = 1024, 200, 41, 1024
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60e9:
        pytest.skip("needs ~45 GB of free HBM")
    from warprnnt_pytorch import warp_rnnt
    g = torch.Generator(device=dev).manual_seed(8)
    x = torch.empty((N, T, U, A), dtype=torch.bfloat16, device=dev)
    for i in range(0, N, 64):                                          # generated slab by slab (no 34 GB fp32 temporary)
        x[i:i + 64] = torch.rand((64, T, U, A), generator=g, device=dev).to(torch.bfloat16)
    assert x.numel() > 2 ** 31
    labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
    tl = torch.randint(T // 2, T + 1, (N,), generator=g, device=dev, dtype=torch.int32)
    ll = torch.randint((U - 1) // 2, U, (N,), generator=g, device=dev, dtype=torch.int32)
    tl[0], ll[0] = T, U - 1
    tl[N - 1], ll[N - 1] = T, U - 1
    costs = torch.zeros(N)
    grads = torch.empty_like(x)
    grads.view(torch.int16).fill_(0x4110)                              # 9.0 in bf16: every element must be overwritten
    assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, costs, grads, 0, 0) == 0
    assert torch.isfinite(costs).all()
    pick = [0, 511, 1023]
    ref_c, ref_g, mag = oracle.rnnt_logits(x[pick].double().cpu().numpy(), labels[pick].cpu().numpy(),
                                           tl[pick].cpu().numpy(), ll[pick].cpu().numpy(), want_mag=True)
    assert np.abs(costs[pick].double().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    got_g = grads[pick].double().cpu().numpy()
    assert np.abs(got_g - ref_g).max() < 4e-3       # absolute (bf16 quantum at |g| ~ 1): cannot see the softmax term here ...
    oracle.assert_grads(got_g, ref_g, mag, torch.bfloat16)             # ... per element: one rounding of the stored value
    # negative control: the non-special columns zeroed pass the absolute bound and FAIL the per-element one
    bad = np.where(mag > np.abs(ref_g) * (1 + 1e-9), got_g, 0.0)
    assert np.abs(bad - ref_g).max() < 4e-3 and not oracle.grad_check(bad, ref_g, mag, torch.bfloat16)["passed"]
    del bad, got_g, mag
    t_idx = torch.arange(T, device=dev).view(1, T, 1)
    u_idx = torch.arange(U, device=dev).view(1, 1, U)
    worst_row, worst_pad = 0.0, 0.0
    for i in range(0, N, 64):
        gs = grads[i:i + 64].float()
        # row sums against the sum of the per-element quanta of the row (the 0.35 this replaces was vacuous)
        worst_row = max(worst_row, (gs.sum(-1).abs() / oracle.rowsum_bound(gs.abs().sum(-1), torch.bfloat16)).max().item())
        pad = (t_idx >= tl[i:i + 64].view(-1, 1, 1)) | (u_idx > ll[i:i + 64].view(-1, 1, 1))
        worst_pad = max(worst_pad, gs.abs().amax(-1)[pad].max().item())
    assert worst_row <= 1.0 and worst_pad == 0.0, (worst_row, worst_pad)
    # forward-only scoring of the same tensor: bit-equal costs
    costs2 = torch.zeros(N)
    assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, costs2, torch.zeros(0, device=dev, dtype=torch.bfloat16), 0, 0) == 0
    assert torch.equal(costs, costs2)


@pytest.mark.parametrize("shape", [(1, 6, 600, 5),       # fp64, 10 wavefronts: lattice_kernel<double,16>
                                   (2, 3, 1024, 4),      # fp64 at the maxU limit, 16 wavefronts
                                   (1, 40, 513, 3),      # first size past the 8-wavefront instantiation
                                   (2, 70, 320, 6)])     # fp64, 5 wavefronts (the c4 lattice width)
def test_wide_lattices_fp64(oracle, shape):
    N, T, U, A = shape
    rng = np.random.default_rng(N * 7 + T * 5 + U * 3 + A)
    acts = rng.standard_normal(shape) * 1.5
    labels = rng.integers(1, A, size=(N, U - 1))
    tl = rng.integers(max(1, T // 2), T + 1, size=N); tl[0] = T
    ll = rng.integers((U - 1) // 2, U, size=N); ll[-1] = U - 1
    ref_c, ref_g = oracle.rnnt_logits(acts, labels, tl, ll)
    c64, g64 = run_gpu(acts, labels, tl, ll, dtype=torch.float64)
    assert np.abs(c64 - ref_c).max() <= 1e-10 * max(1.0, np.abs(ref_c).max())
    assert np.abs(g64 - ref_g).max() < 1e-9
    c32, g32 = run_gpu(acts, labels, tl, ll)
    r32c, r32g, mag = oracle.rnnt_logits(acts.astype(np.float32).astype(np.float64), labels, tl, ll, want_mag=True)
    assert np.abs(c32 - r32c).max() <= 1e-4 * max(1.0, np.abs(r32c).max())
    assert np.abs(g32 - r32g).max() < 5e-4            # ~1000 fp32 lattice steps (north_star: 1e-3)
    oracle.assert_grads(g32, r32g, mag, torch.float32)


def test_batch_size_is_not_limited(oracle):
    """The reference puts the samples on gridDim.x (include/detail/gpu_rnnt.h:127-128) and so takes any batch size;
    here the kernels with the samples on gridDim.y run the batch in slices of 65535.  N on, just past and well past
    that hardware limit: every sample against the oracle.  maxT*maxU >= 2^29 stays INVALID_VALUE, as
    does N > 65535 for the additive-joint extension."""
    from warprnnt_pytorch import _lib
    for N in (65535, 65536, 200000):
        T, U, A = 2, 2, 3
        rng = np.random.default_rng(N)
        acts = rng.standard_normal((N, T, U, A)).astype(np.float32)
        labels = rng.integers(1, A, size=(N, U - 1))
        tl, ll = rng.integers(1, T + 1, size=N), rng.integers(0, U, size=N)
        tl[0], ll[0] = T, U - 1
        ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll)
        costs, grads = run_gpu(acts, labels, tl, ll)
        assert np.abs(costs - ref_c).max() < 1e-4 and np.abs(grads - ref_g).max() < 1e-4, N
    # the wavefront-per-row statistics kernel and the row-form gradient kernel (an unaligned tensor) across the slice edge
    N, T, U, A = 65540, 1, 2, 1100
    rng = np.random.default_rng(3)
    acts = rng.standard_normal((N, T, U, A)).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1))
    tl, ll = np.full(N, T), rng.integers(0, U, size=N)
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll)
    costs, grads = run_gpu(acts, labels, tl, ll)
    assert np.abs(costs - ref_c).max() < 1e-4 and np.abs(grads - ref_g).max() < 1e-4
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    x = torch.zeros(8, device=dev)
    i = torch.ones(8, dtype=torch.int32, device=dev)
    host = torch.zeros(8)
    for kw, n in ((dict(maxT=1 << 20, maxU=512), 1), (dict(maxT=1 << 28, maxU=2), 1)):
        opt = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, batch_first=True, **kw)
        st = lib.compute_rnnt_loss(x.data_ptr(), None, i.data_ptr(), i.data_ptr(), i.data_ptr(), 3, n, host.data_ptr(),
                                   x.data_ptr(), opt)
        assert st == 2, (kw, n, st)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, batch_first=True, maxT=2, maxU=2)
    st = lib.compute_rnnt_loss_add(x.data_ptr(), x.data_ptr(), None, None, i.data_ptr(), i.data_ptr(), i.data_ptr(), 3, 65536,
                                   x.data_ptr(), x.data_ptr(), opt)
    assert st == 2, st


def test_device_side_lengths_are_validated(oracle):
    """T_b > maxT / U_b > maxU / T_b < 1 on the DEVICE: the reference's GPU path reads garbage silently; its CPU
    path has no check either, this library's CPU location returns INVALID_VALUE (csrc/rnnt_cpu.cpp).  The GPU
    location now agrees: compute_rnnt_loss returns INVALID_VALUE (the sample's cost carries a marker NaN, every
    kernel clamps the lengths, nothing is touched out of bounds), the asynchronous entries leave the NaN in the
    device costs, the flagged sample's gradient is all zeros, and the other samples of the batch are unaffected."""
    from warprnnt_pytorch import _lib, warp_rnnt
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    N, T, U, A = 4, 9, 5, 11
    acts = rng.standard_normal((N, T, U, A)).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    good_tl, good_ll = np.array([T, 4, T, 6], np.int32), np.array([U - 1, 2, 0, U - 1], np.int32)
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, good_tl, good_ll)
    x = torch.tensor(acts, device=dev)
    lab = torch.tensor(labels, device=dev)
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, maxT=T, maxU=U, batch_first=True)
    for bad_b, bad_t, bad_l in ((1, T + 1, 2), (3, 6, U), (2, 0, 0), (0, 1 << 30, U - 1), (1, 4, -2)):
        tl, ll = good_tl.copy(), good_ll.copy()
        tl[bad_b], ll[bad_b] = bad_t, bad_l
        ttl, tll = torch.tensor(tl, device=dev), torch.tensor(ll, device=dev)
        grads = torch.zeros_like(x)
        costs = torch.zeros(N)
        st = lib.compute_rnnt_loss(x.data_ptr(), grads.data_ptr(), lab.data_ptr(), tll.data_ptr(), ttl.data_ptr(), A, N,
                                   costs.data_ptr(), ws.data_ptr(), opt)
        assert st == 2, (bad_b, bad_t, bad_l, st)
        dcosts = torch.zeros(N, device=dev)
        warp_rnnt.gpu_rnnt_async(x, lab, ttl, tll, dcosts, grads, 0, workspace=ws)
        torch.cuda.synchronize()
        dc = dcosts.cpu().numpy()
        ok = [b for b in range(N) if b != bad_b]
        assert np.isnan(dc[bad_b]) and np.abs(dc[ok] - ref_c[ok]).max() < 1e-4
        assert np.abs(grads.cpu().numpy()[ok] - ref_g[ok]).max() < 1e-4
        assert not grads[bad_b].any()                  # the flagged sample: zero gradient, not a function of garbage
    # and the valid batch still succeeds afterwards
    costs = torch.zeros(N)
    ttl, tll = torch.tensor(good_tl, device=dev), torch.tensor(good_ll, device=dev)
    grads = torch.zeros_like(x)
    assert lib.compute_rnnt_loss(x.data_ptr(), grads.data_ptr(), lab.data_ptr(), tll.data_ptr(), ttl.data_ptr(), A, N,
                                 costs.data_ptr(), ws.data_ptr(), opt) == 0
    assert np.abs(costs.numpy() - ref_c).max() < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_forward_and_backward_likelihoods_agree(dtype):
    """|llForward - llBackward| over every lattice-kernel form (one wavefront, one and two columns per lane, 3..8
    wavefronts), ragged lengths: the two recursions share nothing but the log-probs, so their agreement checks the
    whole lattice (reference include/detail/cpu_rnnt.h:167-170)."""
    dev = torch.device("cuda:0")
    from warprnnt_pytorch import _lib
    g = torch.Generator(device=dev).manual_seed(17)
    for N, T, U, A in ((5, 40, 33, 7), (3, 300, 64, 5), (3, 90, 130, 4), (2, 700, 257, 3), (2, 50, 301, 6), (2, 30, 700, 3),
                       (2, 20, 1024, 2)):
        x = (torch.randn((N, T, U, A), generator=g, device=dev) * 3.0).to(dtype)
        labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32).cpu().numpy()
        tl = np.array([T] + [max(1, T - 7 * (i + 1)) for i in range(N - 1)], np.int32)
        ll = np.array([U - 1] + [max(0, U - 1 - 11 * (i + 1)) for i in range(N - 1)], np.int32)
        ws = torch.empty(_lib.workspace_bytes(T, U, N, True, x.element_size()), dtype=torch.uint8, device=dev)
        costs, _ = _gpu_call(x, labels, tl, ll, ws)
        llf, llb = _likelihoods(ws, N, T, U, dtype)
        tol = 1e-5 if dtype == torch.float32 else 1e-12
        assert np.abs(llf + costs.double().numpy()).max() <= 2e-7 * np.abs(llf).max(), (N, T, U, A)
        assert np.abs(llf - llb).max() <= tol * np.abs(llf).max(), (N, T, U, A, np.abs(llf - llb).max())


def test_wide_lattice_sweep(oracle):
    """The multi-wavefront lattice forms with ragged batches: 48 seeded problems with maxU between 65 and 1024 (one and two
    columns per lane, every instantiation), T_b and U_b anywhere in range -- U_b on, just before and just after wavefront and
    lane-pair boundaries --, fp32 and fp64, against the oracle."""
    rng = np.random.default_rng(20260926)
    edges = [1, 2, 63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 258, 319, 320, 383, 384, 385, 511, 512, 513,
             514, 639, 640, 641, 767, 768, 769, 1023, 1024]
    for it in range(48):
        U = int(rng.choice([65, 66, 100, 128, 129, 192, 200, 256, 257, 258, 300, 320, 321, 384, 400, 512, 513, 514, 600, 640, 700, 768, 1000, 1024]))
        N = int(rng.integers(2, 5))
        T = int(rng.integers(1, 40))
        A = int(rng.choice([2, 3, 5, 9]))
        acts = rng.standard_normal((N, T, U, A)) * float(rng.choice([0.5, 2.0, 5.0]))
        blank = int(rng.integers(0, A))
        labels = rng.integers(0, A, size=(N, U - 1))
        labels[labels == blank] = (blank + 1) % A
        tl = rng.integers(1, T + 1, size=N); tl[int(rng.integers(0, N))] = T
        cand = [e for e in edges if e <= U]
        ul = np.array([int(rng.choice(cand)) if rng.random() < 0.7 else int(rng.integers(1, U + 1)) for _ in range(N)])
        ul[int(rng.integers(0, N))] = U
        ll = ul - 1
        dtype = torch.float64 if it % 3 == 0 else torch.float32
        x = acts if dtype == torch.float64 else acts.astype(np.float32).astype(np.float64)
        ref_c, ref_g = oracle.rnnt_logits(x, labels, tl, ll, blank)
        costs, grads = run_gpu(x, labels, tl, ll, blank, dtype=dtype)
        # fp32: north_star's 1e-3 -- a thousand lattice steps on logits of magnitude 25 sit right at it (the log-probs
        # themselves are fp32: 2e-6 each); the shorter lattices stay under 5e-4
        tol_c, tol_g = (1e-10, 1e-9) if dtype == torch.float64 else (1e-4, 1e-3 if U >= 512 else 5e-4)
        assert np.abs(costs - ref_c).max() <= tol_c * max(1.0, np.abs(ref_c).max()), (it, N, T, U, A, ul.tolist(), str(dtype))
        assert np.abs(grads - ref_g).max() < tol_g, (it, N, T, U, A, ul.tolist(), str(dtype))
        for b in range(N):
            assert not grads[b, tl[b]:].any() and not grads[b, :, ll[b] + 1:].any()


@pytest.mark.parametrize("case", [((5, 790, 12, 33), torch.float32), ((4, 700, 90, 20), torch.float32), ((3, 400, 400, 9), torch.float32),
                                  ((6, 800, 20, 1024), torch.bfloat16), ((2, 780, 5, 7), torch.float64),
                                  # 2N > compute units: the one-stream schedule takes the log-domain lattice, and so must the halves
                                  # (each of which alone would fit the linear-domain chain's one-block-per-CU rule); odd sample size
                                  ((300, 790, 12, 6), torch.float32), ((5, 771, 3, 3), torch.float32)],
                         ids=lambda c: "x".join(map(str, c[0])) + "-" + str(c[1]).split(".")[-1])
def test_two_half_schedule_on_a_second_stream(oracle, case):
    """rnnt_set_aux_stream: on long lattices the batch is split in two and the lattice kernel of one half runs on the caller's second
    stream beside the other half's streaming kernels.  Same bits as the one-stream schedule (costs, gradients, score-only costs,
    the two-phase pair), ragged lengths and an odd batch included; a NaN logit still poisons its own sample only; capturable."""
    from warprnnt_pytorch import _lib, warp_rnnt
    shape, dtype = case
    N, T, U, A = shape
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(sum(shape))
    acts = torch.tensor(rng.standard_normal(shape).astype(np.float32), device=dev).to(dtype)
    labels = torch.tensor(rng.integers(1, A, size=(N, U - 1)).astype(np.int32), device=dev)
    tl = np.full(N, T, dtype=np.int32); ll = np.full(N, U - 1, dtype=np.int32)
    tl[-1] = T - 7; ll[-1] = max(0, U - 3)
    if N > 2:
        tl[1] = T // 2
    t_tl, t_ll = torch.tensor(tl, device=dev), torch.tensor(ll, device=dev)
    cdt = torch.float64 if dtype == torch.float64 else torch.float32

    def run(want_grad=True, two_phase=False, x=acts):
        costs = torch.zeros(N, dtype=cdt, device=dev)
        grads = torch.full_like(x, 7.0) if want_grad else torch.zeros(0, device=dev, dtype=dtype)
        if two_phase:
            ws = warp_rnnt.gpu_rnnt_fwd(x, labels, t_tl, t_ll, costs, 0, True)
            warp_rnnt.gpu_rnnt_bwd(x, grads, None, ws, 0)
        else:
            warp_rnnt.gpu_rnnt_async(x, labels, t_tl, t_ll, costs, grads, 0)
        torch.cuda.synchronize()
        return costs.cpu().numpy(), grads.float().cpu().numpy()

    modes = (("full", {}), ("score", dict(want_grad=False)), ("two", dict(two_phase=True)))
    lib.rnnt_set_aux_stream(None)
    ref = {k: run(**kw) for k, kw in modes}
    bad = acts.clone(); bad[N - 1, 3, 0, 1] = float("nan")
    ref_bad = run(x=bad)
    side = torch.cuda.Stream(dev)
    warp_rnnt.set_aux_stream(side)
    try:
        got = {k: run(**kw) for k, kw in modes}
        got_bad = run(x=bad)
        # capture: the second stream joins the capture through the fork event
        costs = torch.zeros(N, dtype=cdt, device=dev); grads = torch.zeros_like(acts)
        cap = torch.cuda.Stream(dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            ws = warp_rnnt.gpu_rnnt_async(acts, labels, t_tl, t_ll, costs, grads, 0)       # warm-up outside the capture
            cap.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=cap):
                warp_rnnt.gpu_rnnt_async(acts, labels, t_tl, t_ll, costs, grads, 0, workspace=ws)
        torch.cuda.current_stream(dev).wait_stream(cap)
        costs.zero_(); grads.zero_()
        graph.replay(); torch.cuda.synchronize()
        replay = (costs.cpu().numpy(), grads.float().cpu().numpy())
    finally:
        warp_rnnt.set_aux_stream(None)
    for k in ref:
        assert np.array_equal(got[k][0], ref[k][0]) and np.array_equal(got[k][1], ref[k][1]), k
    assert np.array_equal(replay[0], ref["full"][0]) and np.array_equal(replay[1], ref["full"][1])
    assert np.isnan(got_bad[0][N - 1]) and np.array_equal(got_bad[0][:N - 1], ref_bad[0][:N - 1])
    assert np.array_equal(np.isnan(got_bad[1]), np.isnan(ref_bad[1]))
    # and the one-stream result is the oracle's (first and last sample)
    pick = [0, N - 1]
    rc, rg, mag = oracle.rnnt_logits(acts[pick].double().cpu().numpy(), labels[pick].cpu().numpy(), tl[pick], ll[pick], want_mag=True)
    assert np.abs(got["full"][0][pick] - rc).max() <= 1e-4 * np.abs(rc).max()
    assert np.abs(got["full"][1][pick] - rg).max() <= (1e-3 if dtype != torch.bfloat16 else 4e-3)
    # per element (the absolute bound cannot see a non-special entry at A = 1024); ~800 diagonals of fp32 ahead of the rounding
    # (`got` went through float32 on its way to the host: fp64 storage is judged as fp32 here, its own bound is elsewhere)
    oracle.assert_grads(got["full"][1][pick], rg, mag, dtype if dtype != torch.float64 else torch.float32, rel=1e-3)


def test_aux_stream_can_be_replaced_and_released():
    """rnnt_set_aux_stream again with another stream, and with NULL: the fork / join events are dropped and made again for
    the next call (they used to be created once per thread, on whichever device was current, and never
    destroyed); results stay the one-stream schedule's bits throughout."""
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    N, T, U, A = 4, 800, 9, 16
    rng = np.random.default_rng(5)
    acts = torch.tensor(rng.standard_normal((N, T, U, A)).astype(np.float32), device=dev)
    labels = torch.tensor(rng.integers(1, A, size=(N, U - 1)).astype(np.int32), device=dev)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev); ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)

    def run():
        costs = torch.zeros(N, device=dev); grads = torch.empty_like(acts)
        warp_rnnt.gpu_rnnt_async(acts, labels, tl, ll, costs, grads, 0)
        torch.cuda.synchronize()
        return costs.cpu().numpy(), grads.cpu().numpy()

    warp_rnnt.set_aux_stream(None)
    ref = run()
    try:
        for _ in range(3):
            side = torch.cuda.Stream(dev)
            warp_rnnt.set_aux_stream(side)
            for _ in range(2):
                got = run()
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
            warp_rnnt.set_aux_stream(None)
            got = run()
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    finally:
        warp_rnnt.set_aux_stream(None)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_aux_stream_follows_the_thread_to_another_device():
    """A thread that hands over a stream of ANOTHER GPU after having used the schedule on the first: the events are made
    again on the device of the call."""
    from warprnnt_pytorch import warp_rnnt
    N, T, U, A = 4, 800, 9, 16
    rng = np.random.default_rng(6)
    x = rng.standard_normal((N, T, U, A)).astype(np.float32)
    lab = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    outs = []
    try:
        for d in (0, 1, 0):
            dev = torch.device("cuda", d)
            with torch.cuda.device(dev):
                acts = torch.tensor(x, device=dev); labels = torch.tensor(lab, device=dev)
                tl = torch.full((N,), T, dtype=torch.int32, device=dev); ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
                warp_rnnt.set_aux_stream(torch.cuda.Stream(dev))
                costs = torch.zeros(N, device=dev); grads = torch.empty_like(acts)
                warp_rnnt.gpu_rnnt_async(acts, labels, tl, ll, costs, grads, 0)
                torch.cuda.synchronize(dev)
                outs.append((costs.cpu().numpy(), grads.cpu().numpy()))
    finally:
        warp_rnnt.set_aux_stream(None)
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
