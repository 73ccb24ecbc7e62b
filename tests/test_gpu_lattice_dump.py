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
