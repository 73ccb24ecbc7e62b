"""-m gpu: compute_rnnt_loss_lattice_dump -- the alpha / beta tables of one sample out of the workspace, the counterpart of the
reference's -DDEBUG_KERNEL dumps (include/detail/gpu_rnnt.h:136-156,175-191) -- against a plain numpy forward / backward pass
(the recursions of include/detail/cpu_rnnt.h:175-251) in fp64.  Every lattice kernel form: the linear-domain chain (U <= 64,
few samples), the one-wavefront log-domain form (many samples), multi-wavefront blocks (U = 130), two columns per lane (U = 300)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def numpy_lattice(lp, labels, T, U, blank):
    """alpha, beta (T, U) in natural logs from log-probs lp (T, U, A)."""
    a = np.full((T, U), -np.inf); b = np.full((T, U), -np.inf)
    a[0, 0] = 0.0
    for t in range(T):
        for u in range(U):
            if t > 0:
                a[t, u] = np.logaddexp(a[t, u], a[t - 1, u] + lp[t - 1, u, blank])
            if u > 0:
                a[t, u] = np.logaddexp(a[t, u], a[t, u - 1] + lp[t, u - 1, labels[u - 1]])
    b[T - 1, U - 1] = lp[T - 1, U - 1, blank]
    for t in range(T - 1, -1, -1):
        for u in range(U - 1, -1, -1):
            if t < T - 1:
                b[t, u] = np.logaddexp(b[t, u], b[t + 1, u] + lp[t, u, blank])
            if u < U - 1:
                b[t, u] = np.logaddexp(b[t, u], b[t, u + 1] + lp[t, u, labels[u]])
    return a, b


@pytest.mark.parametrize("shape,dtype", [((3, 12, 6, 9), torch.float32), ((3, 12, 6, 9), torch.float64), ((300, 9, 5, 4), torch.float32),
                                         ((2, 20, 130, 5), torch.float32), ((2, 30, 300, 4), torch.float32), ((2, 30, 300, 4), torch.float64),
                                         ((2, 9, 5, 16), torch.bfloat16)])
def test_dump_equals_a_numpy_forward_backward(shape, dtype):
    from warprnnt_pytorch import _lib, warp_rnnt
    lib = _lib.lib()
    N, T, U, A = shape
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(sum(shape))
    blank = int(rng.integers(0, A))
    x = torch.tensor(rng.standard_normal(shape), device=dev).to(dtype)
    labels = rng.integers(0, A, size=(N, U - 1)).astype(np.int32)
    tl = rng.integers(max(1, T // 2), T + 1, size=N).astype(np.int32); tl[0] = T
    ll = rng.integers((U - 1) // 2, U, size=N).astype(np.int32); ll[0] = U - 1
    t_lab, t_tl, t_ll = (torch.tensor(v, device=dev) for v in (labels, tl, ll))
    cdt = torch.float64 if dtype == torch.float64 else torch.float32
    costs = torch.zeros(N, dtype=cdt, device=dev)
    grads = torch.empty_like(x)
    ws = warp_rnnt.gpu_rnnt_async(x, t_lab, t_tl, t_ll, costs, grads, blank)
    code = {torch.float32: _lib.DT_F32, torch.float64: _lib.DT_F64, torch.bfloat16: _lib.DT_BF16}[dtype]
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=blank, maxT=T, maxU=U,
                           batch_first=True)
    for b in sorted({0, N - 1}):
        a_out = torch.zeros(T * U, dtype=torch.float64, device=dev); b_out = torch.zeros_like(a_out)
        st = lib.compute_rnnt_loss_lattice_dump(ws.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(), N, b, opt, code, a_out.data_ptr(), b_out.data_ptr())
        assert st == 0
        torch.cuda.synchronize()
        Tb, Ub = int(tl[b]), int(ll[b]) + 1
        lp = torch.log_softmax(x[b, :Tb, :Ub].double().cpu(), -1).numpy()
        ra, rb = numpy_lattice(lp, labels[b], Tb, Ub, blank)
        ga, gb = a_out.view(T, U).cpu().numpy(), b_out.view(T, U).cpu().numpy()
        tol = 1e-9 if dtype == torch.float64 else 2e-4 * max(1.0, np.abs(ra).max() / 30)      # fp32 lattice values, re-centred per chunk
        assert np.abs(ga[:Tb, :Ub] - ra).max() <= tol and np.abs(gb[:Tb, :Ub] - rb).max() <= tol
        assert np.isnan(ga[Tb:]).all() and np.isnan(ga[:, Ub:]).all() and np.isnan(gb[Tb:]).all()      # outside the lattice: NaN
        assert abs(-gb[0, 0] - float(costs[b])) <= 1e-4 * max(1.0, abs(float(costs[b])))                 # beta(0,0) = log P(y|x)
    bad = lib.compute_rnnt_loss_lattice_dump(ws.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(), N, N, opt, code, a_out.data_ptr(), b_out.data_ptr())
    assert bad == 2                                                                                     # sample outside the batch


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("shape", [(40, 300, 200, 2), (16, 800, 200, 2)])
def test_record_table_overlays_consumed_lattice_blocks(oracle, shape, packed):
    """Workspace layout of round 6 (rnnt_host.h, make_layout): a record table beyond 32 MB is written group by group over the
    lattice blocks of the samples already consumed.  (a) every sample's gradient is still the oracle's -- nothing live was
    overwritten, padded and packed layout, one-stream and (T + U >= 768, second stream) two-half schedule; (b) the lattice dump
    says which samples lost their alpha / beta (all-NaN) and still returns the others; (c) after a score-only call every
    sample's alpha is there; (d) the workspace is smaller than records + blocks."""
    from warprnnt_pytorch import _lib, warp_rnnt
    from warprnnt_pytorch.packed import pack_joint, row_offsets
    lib = _lib.lib()
    N, T, U, A = shape
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(sum(shape))
    x = torch.tensor(rng.standard_normal(shape).astype(np.float32), device=dev)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl = rng.integers(T // 2, T + 1, size=N).astype(np.int32); tl[0] = T; tl[N - 1] = T
    ll = rng.integers((U - 1) // 2, U, size=N).astype(np.int32); ll[0] = U - 1; ll[N - 1] = U - 1
    t_lab, t_tl, t_ll = (torch.tensor(v, device=dev) for v in (labels, tl, ll))
    nbytes = _lib.workspace_bytes(T, U, N, True, 4)
    records = 16 * N * T * U
    block = 20 * (T + U - 1 + 32) * ((U + 7) // 8 * 8)
    assert records > (32 << 20) and nbytes < records + N * block                                       # (d): the table's tail lies over blocks
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ws.fill_(0xA5)
    ref_c, ref_g, mag = oracle.rnnt_logits(x.double().cpu().numpy(), labels, tl, ll, want_mag=True)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=0, maxT=T, maxU=U, batch_first=True)
    aux = torch.cuda.Stream(dev)
    for use_aux in ((False, True) if not packed else (False,)):
        if use_aux:
            warp_rnnt.set_aux_stream(aux)
        try:
            costs = torch.zeros(N, device=dev)
            if packed:
                p = pack_joint(x, t_tl, t_ll).contiguous()
                offs = row_offsets(t_tl, t_ll)
                g = torch.full_like(p, 7.0)
                scale = torch.ones(N, device=dev)
                st = lib.compute_rnnt_loss_packed(p.data_ptr(), g.data_ptr(), t_lab.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(), offs.data_ptr(),
                                                  p.shape[0], A, N, costs.data_ptr(), scale.data_ptr(), ws.data_ptr(), opt, _lib.DT_F32, 0.0)
                assert st == 0
                torch.cuda.synchronize()
                got = torch.zeros_like(x)
                o = offs.cpu().numpy()
                for b in range(N):
                    got[b, :tl[b], :ll[b] + 1] = g[o[b]:o[b + 1]].view(int(tl[b]), int(ll[b]) + 1, A)
            else:
                got = torch.full_like(x, 7.0)
                warp_rnnt.gpu_rnnt_async(x, t_lab, t_tl, t_ll, costs, got, 0, workspace=ws)
                torch.cuda.synchronize()
        finally:
            if use_aux:
                warp_rnnt.set_aux_stream(None)
        assert np.abs(costs.double().cpu().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
        oracle.assert_grads(got.double().cpu().numpy(), ref_g, mag, torch.float32, what=("aux" if use_aux else "one stream", packed))   # (a)
    # (b) the dump after the gradient-computing call
    a_out = torch.zeros(T * U, dtype=torch.float64, device=dev); b_out = torch.zeros_like(a_out)

    def dump(b):
        assert lib.compute_rnnt_loss_lattice_dump(ws.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(), N, b, opt, _lib.DT_F32, a_out.data_ptr(), b_out.data_ptr()) == 0
        torch.cuda.synchronize()
        return a_out.view(T, U).cpu().numpy().copy(), b_out.view(T, U).cpu().numpy().copy()
    a0, b0 = dump(0)
    assert np.isnan(a0).all() and np.isnan(b0).all()                           # sample 0: overlaid by the records
    aN, bN = dump(N - 1)
    assert np.isfinite(aN).all() and abs(-bN[0, 0] - ref_c[N - 1]) <= 1e-4 * abs(ref_c[N - 1])
    llf, llb = np.zeros(N), np.zeros(N)
    assert lib.compute_rnnt_loss_likelihoods(ws.data_ptr(), N, opt, _lib.DT_F32, llf.ctypes.data, llb.ctypes.data) == 0
    assert np.abs(llf + ref_c).max() <= 1e-4 * np.abs(ref_c).max() and np.abs(llf - llb).max() <= 1e-5 * np.abs(llf).max()   # kept for EVERY sample
    # (c) a score-only call: every sample's alpha is there again
    if not packed:
        costs2 = torch.zeros(N, device=dev)
        warp_rnnt.gpu_rnnt_async(x, t_lab, t_tl, t_ll, costs2, torch.zeros(0, device=dev), 0, workspace=ws)
        torch.cuda.synchronize()
        a0, _ = dump(0)
        Tb, Ub = int(tl[0]), int(ll[0]) + 1
        lp = torch.log_softmax(x[0, Tb - 1, Ub - 1].double().cpu(), -1).numpy()
        assert np.isfinite(a0[:Tb, :Ub]).all() and abs(-(a0[Tb - 1, Ub - 1] + lp[0]) - ref_c[0]) <= 1e-4 * abs(ref_c[0])
