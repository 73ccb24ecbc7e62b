"""-m gpu: the HIP path (through the C-ABI of include/rnnt.h) against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_gpu(acts_np, labels, act_lens, label_lens, blank=0, dtype=torch.float32, want_grad=True):
    """Raw C-ABI call (warp_rnnt.gpu_rnnt == compute_rnnt_loss[_fp64|_bf16|_fp16], RNNT_GPU)."""
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    x = torch.tensor(acts_np, device=dev).to(dtype).contiguous()
    lab = torch.tensor(np.asarray(labels, dtype=np.int32), device=dev)
    if lab.dim() == 1:
        lab = lab.view(x.shape[0], -1)
    if lab.numel() == 0:      # maxU == 1: no labels exist, but the C-ABI rejects a NULL pointer (as the reference)
        lab = torch.zeros(1, dtype=torch.int32, device=dev)
    tl = torch.tensor(np.asarray(act_lens, dtype=np.int32), device=dev)
    ll = torch.tensor(np.asarray(label_lens, dtype=np.int32), device=dev)
    cost_dtype = dtype if dtype in (torch.float32, torch.float64) else torch.float32
    costs = torch.zeros(x.shape[0], dtype=cost_dtype)
    grads = torch.full_like(x, 123.0) if want_grad else torch.zeros(0, device=dev, dtype=dtype)
    assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, blank, 0) == 0
    torch.cuda.synchronize()
    return costs.numpy().astype(np.float64), (grads.float().cpu().numpy().astype(np.float64)
                                              if want_grad and dtype != torch.float64
                                              else grads.cpu().numpy() if want_grad else None)


SMALL = np.array([[[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
                   [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]])
SMALL_GRADS = np.array([[[[-0.13116688, -0.3999269, 0.17703125, 0.17703125, 0.17703125],
                          [-0.18572757, 0.12247056, -0.18168412, 0.12247056, 0.12247056],
                          [-0.32091254, 0.06269141, 0.06928472, 0.12624499, 0.06269141]],
                         [[0.05456069, -0.21824276, 0.05456069, 0.05456069, 0.05456069],
                          [0.12073959, 0.12073959, -0.48295835, 0.12073959, 0.12073959],
                          [-0.6925882, 0.16871116, 0.18645467, 0.16871116, 0.16871116]]]])


def test_small_golden():
    # reference tests/test_gpu.cu:29-32 (cost 4.495666 +- 1e-4), pytorch_binding/test/test.py:61-78
    costs, grads = run_gpu(SMALL, [[1, 2]], [2], [2])
    assert abs(costs[0] - 4.495666) < 1e-4
    assert np.abs(grads - SMALL_GRADS).max() < 1e-5


@pytest.mark.parametrize("shape", [(3, 17, 6, 40), (2, 50, 10, 15), (5, 9, 6, 7), (2, 33, 70, 12),
                                   (2, 20, 5, 1000), (1, 7, 3, 5003), (2, 12, 130, 9)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_random_vs_oracle(oracle, shape, dtype):
    N, T, U, A = shape
    rng = np.random.default_rng(sum(shape))
    acts = rng.standard_normal(shape) * 2.0
    labels = rng.integers(0, A, size=(N, U - 1))
    blank = int(rng.integers(0, A))
    labels[labels == blank] = (blank + 1) % A
    act_lens = rng.integers(1, T + 1, size=N); act_lens[0] = T
    label_lens = rng.integers(0, U, size=N); label_lens[-1] = U - 1
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, act_lens, label_lens, blank)
    costs, grads = run_gpu(acts, labels, act_lens, label_lens, blank, dtype)
    tol_c, tol_g = (1e-4, 1e-4) if dtype == torch.float32 else (1e-9, 1e-9)
    if dtype == torch.float32:   # inputs were rounded to fp32 before the kernel saw them
        ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float32).astype(np.float64), labels, act_lens,
                                          label_lens, blank)
    assert np.abs(costs - ref_c).max() <= tol_c * max(1.0, np.abs(ref_c).max())
    assert np.abs(grads - ref_g).max() <= tol_g
    # padded region must come back exactly zero (no memset is done by the caller)
    for b in range(N):
        assert not grads[b, act_lens[b]:].any() and not grads[b, :, label_lens[b] + 1:].any()


# ----------------------------------------------------------------------------------------------
# reference golden vectors and reference-generated fixtures
from tests.golden import literals as G                      # noqa: E402
from tests.golden.make_golden import CASES, case_inputs     # noqa: E402

FIX = np.load(__file__.replace("test_gpu_parity.py", "golden/ref_cases.npz"))


def test_options_test_golden():
    # tests/test_gpu.cu:100-200: costs and dense logit grads within 1e-4
    costs, grads = run_gpu(G.OPTIONS_ACTS_6DP, G.OPTIONS_LABELS, [4, 4], [2, 2])
    assert np.abs(costs - G.OPTIONS_COSTS).max() < 1e-4
    assert np.abs(grads - G.OPTIONS_LOGIT_GRADS_6DP).max() < 1e-4


def test_big_test_fp64():
    costs, grads = run_gpu(G.BIG_ACTS, G.OPTIONS_LABELS, [4, 4], [2, 2], dtype=torch.float64)
    assert np.abs(costs - G.OPTIONS_COSTS).max() < 1e-9
    assert np.allclose(grads, G.BIG_GRADS, rtol=1e-3, atol=1e-7)   # literals are fp32-born (test.py:160)


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_fixture(name):
    """Outputs recorded from the reference's own CPU library (fp64) -- variable lengths, blank != 0,
    U_b = 1, T_b = 1, wide U, wide A, and the reference's inf_test / grad_check inputs."""
    acts, labels, tl, ll, blank = case_inputs(name)
    costs, grads = run_gpu(acts, labels, tl, ll, blank)
    ref_c = FIX[name + "/costs64"]
    assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())   # relative 1e-4
    assert np.abs(grads - FIX[name + "/logitgrad64"]).max() < 1e-4
    assert np.isfinite(costs).all() and not np.isnan(grads).any()                  # inf_test
    c64, g64 = run_gpu(acts, labels, tl, ll, blank, dtype=torch.float64)
    assert np.abs(c64 - ref_c).max() <= 1e-10 * max(1.0, np.abs(ref_c).max())
    assert np.abs(g64 - FIX[name + "/logitgrad64"]).max() < 1e-6                  # fixture stored as fp32


def test_forward_only_matches_training_costs(oracle):
    acts, labels, tl, ll, blank = case_inputs("var_a40")
    c_train, _ = run_gpu(acts, labels, tl, ll, blank)
    c_score, g = run_gpu(acts, labels, tl, ll, blank, want_grad=False)            # gradients == NULL
    assert g is None and np.array_equal(c_train, c_score)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_precision_storage(oracle, dtype):
    """16-bit activations/gradients (extension): the kernel sees the rounded inputs, computes in
    fp32 and rounds the gradient once on store: per element one rounding of the stored value
    (oracle.grad_bound: 2^-8 |ref| for bf16, 2^-11 |ref| for fp16, + the fp32 arithmetic ahead of it)."""
    rng = np.random.default_rng(3)
    N, T, U, A = 3, 21, 8, 264
    acts = torch.tensor(rng.standard_normal((N, T, U, A)) * 1.5).to(dtype)
    labels = rng.integers(1, A, size=(N, U - 1))
    tl, ll = np.array([T, 9, T]), np.array([U - 1, U - 1, 3])
    ref_c, ref_g, mag = oracle.rnnt_logits(acts.double().numpy(), labels, tl, ll, want_mag=True)
    costs, grads = run_gpu(acts.double().numpy(), labels, tl, ll, dtype=dtype)
    assert np.abs(costs - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
    oracle.assert_grads(grads, ref_g, mag, dtype)


def test_unaligned_rows_and_odd_vocab(oracle):
    """A not a multiple of the 16-byte packet (rows start at every phase), and a tensor whose base
    pointer is offset by one element."""
    from warprnnt_pytorch import warp_rnnt
    rng = np.random.default_rng(11)
    for A in (3, 5, 50, 257, 1001):
        N, T, U = 2, 7, 4
        acts = rng.standard_normal((N, T, U, A))
        labels = rng.integers(1, A, size=(N, U - 1))
        ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float32).astype(np.float64), labels, [T, T], [U - 1, U - 1])
        costs, grads = run_gpu(acts, labels, [T, T], [U - 1, U - 1])
        assert np.abs(costs - ref_c).max() < 1e-4 and np.abs(grads - ref_g).max() < 1e-5
    dev = torch.device("cuda:0")
    N, T, U, A = 2, 5, 3, 64
    acts = rng.standard_normal((N, T, U, A)).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    buf = torch.zeros(acts.size + 1, device=dev)
    x = buf[1:].view(N, T, U, A)
    x.copy_(torch.tensor(acts))
    gbuf = torch.zeros(acts.size + 3, device=dev)
    g = gbuf[3:].view(N, T, U, A)                           # different phase than acts -> scalar path
    costs = torch.zeros(N)
    warp_rnnt.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor([T, T], dtype=torch.int32, device=dev),
                       torch.tensor([U - 1, U - 1], dtype=torch.int32, device=dev), costs, g, 0, 0)
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, [T, T], [U - 1, U - 1])
    assert np.abs(costs.numpy() - ref_c).max() < 1e-4 and np.abs(g.cpu().numpy() - ref_g).max() < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("A", [29, 511, 1023, 1025, 1279, 2047, 2049, 4097])
def test_vocabularies_next_to_powers_of_two(oracle, A, dtype):
    """V + blank and V - 1 symbols: rows that start at every 2- / 4-byte phase.  Covers the statistics kernels' forms for
    such rows -- aligned 8-byte LDS words with masked edges (rows up to 4 KB), covering 16-byte packets with masked edges
    and the fifth packet of a 2^k + 1 row (longer rows) -- and the straddling packets of the gradient stream."""
    from warprnnt_pytorch import warp_rnnt
    rng = np.random.default_rng(A)
    N, T, U = 3, 6, 4
    dev = torch.device("cuda:0")
    x = torch.tensor(rng.standard_normal((N, T, U, A)) * 2.0, dtype=dtype, device=dev)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl, ll = np.array([T, T - 2, 3], dtype=np.int32), np.array([U - 1, 1, 0], dtype=np.int32)
    ref_c, ref_g, mag = oracle.rnnt_logits(x.double().cpu().numpy(), labels, tl, ll, want_mag=True)
    costs, grads = torch.zeros(N), torch.empty_like(x)
    assert warp_rnnt.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev), torch.tensor(ll, device=dev),
                              costs, grads, 0, 0) == 0
    assert np.abs(costs.numpy() - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    if dtype == torch.float32:
        assert np.abs(grads.double().cpu().numpy() - ref_g).max() <= 1e-5
    oracle.assert_grads(grads.double().cpu().numpy(), ref_g, mag, dtype)   # per element: one rounding of the stored value
    assert not grads[1, T - 2:].any() and not grads[2, :, 1:].any()                             # padding: exact zeros


def test_gpu_status_codes():
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    x = torch.zeros(1, 2, 3, 4, device=dev)
    i = torch.ones(4, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.workspace_bytes(2, 3, 1, True, 4), dtype=torch.uint8, device=dev)
    costs = torch.zeros(1)

    def call(**kw):
        o = dict(loc=1, num_threads=0, stream=None, blank_label=0, maxT=2, maxU=3, batch_first=True)
        o.update(kw)
        return lib.compute_rnnt_loss(x.data_ptr(), None, i.data_ptr(), i.data_ptr(), i.data_ptr(), 4, 1,
                                     costs.data_ptr(), ws.data_ptr(), _lib.rnntOptions(**o))
    assert call(blank_label=4) == 2 and call(blank_label=-1) == 2 and call(maxU=2000) == 2
    torch.cuda.synchronize()


def test_async_entry_and_fused_grad_scale(oracle):
    """compute_rnnt_loss_async: device costs, no sync, per-sample gradient scale folded into the
    gradient kernel (what autograd's grads.mul_(grad_output) does in the reference binding)."""
    from warprnnt_pytorch import warp_rnnt
    acts, labels, tl, ll, blank = case_inputs("blank5_a19")
    dev = torch.device("cuda:0")
    x = torch.tensor(acts, dtype=torch.float32, device=dev)
    args = [torch.tensor(a, device=dev) for a in (labels, tl, ll)]
    costs = torch.empty(x.shape[0], device=dev)
    grads = torch.empty_like(x)
    scale = torch.tensor([0.5, -2.0, 0.0, 3.0], device=dev)
    ws = warp_rnnt.gpu_rnnt_async(x, *args, costs, grads, blank, grad_scale=scale)
    torch.cuda.synchronize()
    del ws
    ref_c = FIX["blank5_a19/costs64"]
    ref_g = FIX["blank5_a19/logitgrad64"] * scale.cpu().numpy()[:, None, None, None]
    assert np.abs(costs.cpu().numpy() - ref_c).max() < 1e-4 * np.abs(ref_c).max()
    assert np.abs(grads.cpu().numpy() - ref_g).max() < 3e-4


@pytest.mark.parametrize("loader", ["ext", "ctypes"])
@pytest.mark.parametrize("async_entry", [True, False])
def test_pytorch_binding_on_gpu(monkeypatch, async_entry, loader):
    """pytorch_binding/test/test.py on the device: CPU and GPU must give identical grads.  Both
    routes of the wrapper: the asynchronous entry (device costs) and the reference's host-costs
    entry compute_rnnt_loss -- and both bindings: the compiled extension module (csrc/binding.cpp, the
    reference's form: pytorch_binding/src/binding.cpp) and the ctypes loader."""
    import warprnnt_pytorch
    from warprnnt_pytorch import RNNTLoss, warp_rnnt
    if loader == "ext":
        assert warp_rnnt.binding() == "ext", "the compiled extension module was not built (python -c 'import __graft_entry__ as g; g.build()')"
    else:
        monkeypatch.setattr(warp_rnnt, "_EXT", None)
        assert warp_rnnt.binding() == "ctypes"
    monkeypatch.setattr(warprnnt_pytorch, "_ASYNC_GPU", async_entry)
    dev = torch.device("cuda:0")
    for acts_np, labels, cost, grads_ref in ((G.SMALL_ACTS, [[1, 2]], G.SMALL_COST, G.SMALL_GRADS),
                                             (G.BIG_ACTS, [[1, 2], [1, 1]], sum(G.OPTIONS_COSTS), G.BIG_GRADS)):
        x = torch.tensor(acts_np, dtype=torch.float32, device=dev, requires_grad=True)
        n = x.shape[0]
        lab = torch.tensor(labels, dtype=torch.int32, device=dev)
        tl = torch.full((n,), x.shape[1], dtype=torch.int32, device=dev)
        ll = torch.tensor([len(l) for l in labels], dtype=torch.int32, device=dev)
        loss = RNNTLoss(reduction='sum')(x, lab, tl, ll)
        assert loss.is_cuda and loss.shape == (1,)
        loss.sum().backward()
        assert np.allclose(loss.item(), cost, rtol=1e-5)
        assert np.allclose(x.grad.cpu().numpy(), grads_ref, rtol=1e-3, atol=1e-6)
        m = RNNTLoss(reduction='mean')(x.detach().requires_grad_(True), lab, tl, ll)
        assert np.allclose(m.item(), cost / n, rtol=1e-5)
        # per-sample losses with a per-sample grad_output, the reference's validation errors, and a no-grad call
        x2 = x.detach().clone().requires_grad_(True)
        per = RNNTLoss(reduction='none')(x2, lab, tl, ll)
        w = torch.arange(1, n + 1, dtype=per.dtype, device=dev)
        (per * w).sum().backward()
        assert np.allclose(x2.grad.cpu().numpy(), grads_ref * w.cpu().numpy()[:, None, None, None], rtol=1e-3, atol=1e-6)
        with pytest.raises(ValueError, match="Input length mismatch"):
            RNNTLoss()(x, lab, tl - 1, ll)
        with pytest.raises(ValueError, match="Output length mismatch"):
            RNNTLoss()(x, lab, tl, ll - 1)
        with pytest.raises(TypeError, match="labels must be"):
            RNNTLoss()(x, lab.long(), tl, ll)
        with pytest.raises(ValueError, match="must be contiguous"):
            RNNTLoss()(x.transpose(1, 2).contiguous().transpose(1, 2), lab, tl, ll)
        with torch.no_grad():
            assert np.allclose(RNNTLoss(reduction='sum')(x, lab, tl, ll).item(), cost, rtol=1e-5)
        assert RNNTLoss(validate=False)(x, lab, tl, ll).shape == (1,)


def test_pinned_host_costs_are_written_directly():
    """Host costs in pinned memory take the direct-write route (no staged D2H copy): same values, bit for bit, as
    pageable costs; an invalid device-side length is still reported."""
    from warprnnt_pytorch import warp_rnnt
    acts, labels, tl, ll, blank = case_inputs("var_a40")
    dev = torch.device("cuda:0")
    x = torch.tensor(acts, dtype=torch.float32, device=dev)
    lab, tlen, llen = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    outs = []
    for pinned in (False, True):
        costs = torch.full((x.shape[0],), 7.0, pin_memory=pinned)
        grads = torch.zeros_like(x)
        assert warp_rnnt.gpu_rnnt(x, lab, tlen, llen, costs, grads, blank, 0) == 0
        outs.append((costs.clone(), grads.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert np.abs(outs[1][0].numpy() - FIX["var_a40/costs64"]).max() <= 1e-4 * np.abs(FIX["var_a40/costs64"]).max()
    bad = tlen.clone(); bad[0] = x.shape[1] + 3
    with pytest.raises(RuntimeError, match="invalid value"):
        warp_rnnt.gpu_rnnt(x, lab, bad, llen, torch.zeros(x.shape[0], pin_memory=True), torch.zeros_like(x), blank, 0)


def _pageable_call(n, dtype, g):
    """One synchronous C-ABI call with PAGEABLE host costs (guard words around them) against the device-cost entry."""
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    x = torch.randn((n, 6, 4, 9), generator=g, device=dev, dtype=dtype)
    lab = torch.randint(1, 9, (n, 3), generator=g, device=dev, dtype=torch.int32)
    tl = torch.full((n,), 6, dtype=torch.int32, device=dev)
    ll = torch.full((n,), 3, dtype=torch.int32, device=dev)
    host = torch.full((n + 2,), -7.0, dtype=dtype)                 # guard words on both sides
    grads = torch.zeros_like(x)
    assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, host[1:n + 1], grads, 0, 0) == 0
    ref = torch.empty(n, dtype=dtype, device=dev)
    ws = warp_rnnt.gpu_rnnt_fwd(x, lab, tl, ll, ref, 0, False)
    torch.cuda.synchronize()
    del ws
    assert host[0].item() == -7.0 and host[n + 1].item() == -7.0
    assert torch.equal(host[1:n + 1], ref.cpu())
    assert bool(torch.isfinite(host).all()) and bool((host[1:n + 1] > 0).all())


def test_library_makes_no_allocations():
    """north_star: "the workspace sizing / no-internal-malloc contract is preserved" (reference README.md:36-37).
    With the default settings the drop-in call allocates NOTHING: the library's own pinned-byte counter stays 0 and the
    device's free memory (hipMemGetInfo through torch) is the same before and after calls of several shapes, dtypes and
    entry points -- pageable costs take the copy the reference takes (include/detail/gpu_rnnt.h:208-213), pinned costs are
    written directly."""
    from warprnnt_pytorch import _lib, warp_rnnt
    lib = _lib.lib()
    assert lib.get_warprnnt_extension_version() >= 3
    assert lib.rnnt_host_staging(-1) == 0 and lib.rnnt_host_staging_bytes() == 0          # off unless asked for
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    cases = []
    for n, T, U, A, dtype in ((3, 6, 4, 9, torch.float32), (700, 5, 3, 40, torch.float64), (5, 30, 70, 12, torch.float32),
                              (2, 9, 5, 3000, torch.bfloat16)):
        x = torch.randn((n, T, U, A), generator=g, device=dev, dtype=torch.float32).to(dtype)
        lab = torch.randint(1, A, (n, U - 1), generator=g, device=dev, dtype=torch.int32)
        tl = torch.full((n,), T, dtype=torch.int32, device=dev)
        ll = torch.full((n,), U - 1, dtype=torch.int32, device=dev)
        cdt = dtype if dtype == torch.float64 else torch.float32
        ws = torch.empty(_lib.workspace_bytes(T, U, n, True, x.element_size()), dtype=torch.uint8, device=dev)
        cases.append((x, lab, tl, ll, torch.zeros(n, dtype=cdt), torch.zeros(n, dtype=cdt, pin_memory=True),
                      torch.zeros(n, dtype=cdt, device=dev), torch.zeros_like(x), ws))
    torch.cuda.synchronize()

    def sweep():
        for x, lab, tl, ll, pageable, pinned, dcosts, grads, ws in cases:
            assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, pageable, grads, 0, 0, workspace=ws) == 0
            assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, pinned, grads, 0, 0, workspace=ws) == 0
            warp_rnnt.gpu_rnnt_async(x, lab, tl, ll, dcosts, grads, 0, workspace=ws)
            torch.cuda.synchronize()
            assert torch.equal(pageable, pinned) and torch.equal(pinned, dcosts.cpu())
    sweep()                                                # first calls: code objects load (the runtime's, not ours)
    sweep()                                                # (and whatever the runtime sets up lazily for its own copies)
    free = []
    for _ in range(3):
        sweep()
        free.append(torch.cuda.mem_get_info(dev)[0])
    assert free[0] == free[1] == free[2], free             # steady state: not a byte of device memory per call
    assert lib.rnnt_host_staging_bytes() == 0


def test_opt_in_host_staging():
    """rnnt_host_staging(1): pageable host costs come back through the calling thread's pinned staging buffer -- a small
    batch first, then one that makes the buffer grow (fp64: 8 bytes per sample), the same from a second thread (its
    own buffer), a batch past the 1 MB cap (the copy route), always the values of the device-cost entry and untouched
    memory around them; the bytes are counted and rnnt_host_staging_release() returns them."""
    import threading
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    assert lib.rnnt_host_staging(1) == 0
    try:
        assert lib.rnnt_host_staging(-1) == 1
        _pageable_call(3, torch.float32, g)
        assert lib.rnnt_host_staging_bytes() == 4096
        _pageable_call(2000, torch.float64, g)                         # 16 000 bytes: past the first 4 KB buffer
        assert lib.rnnt_host_staging_bytes() == 16384
        _pageable_call(5, torch.float32, g)
        assert lib.rnnt_host_staging_bytes() == 16384
        errs = []

        def worker():
            try:
                torch.cuda.set_device(0)
                _pageable_call(1500, torch.float32, g)
                _pageable_call(7, torch.float64, g)
            except Exception as e:                                     # noqa: BLE001 -- reported to the main thread
                errs.append(e)
        th = threading.Thread(target=worker)
        th.start(); th.join()
        assert not errs, errs
        assert lib.rnnt_host_staging_bytes() == 16384 + 8192           # the second thread's own buffer
        _pageable_call(140000, torch.float64, g)                       # 1.12 MB > cap: copied as by default
        assert lib.rnnt_host_staging_bytes() == 16384 + 8192
        assert lib.rnnt_host_staging_release() == 16384 + 8192 and lib.rnnt_host_staging_bytes() == 0
        _pageable_call(9, torch.float32, g)                            # and it comes back on demand
        assert lib.rnnt_host_staging_bytes() == 4096
    finally:
        lib.rnnt_host_staging(0)
        lib.rnnt_host_staging_release()
    assert lib.rnnt_host_staging_bytes() == 0
    _pageable_call(4, torch.float32, g)                                # off again: nothing allocated
    assert lib.rnnt_host_staging_bytes() == 0


def test_concurrent_threads_on_their_own_streams():
    """Four host threads call the synchronous entry at the same time, each on its own stream and its own shape: every
    call returns exactly what the same call returns alone (no shared mutable state in the library besides the per-thread
    staging buffer)."""
    import threading
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    shapes = [(5, 40, 9, 33), (3, 70, 70, 12), (8, 25, 5, 300), (2, 130, 20, 64)]
    g = torch.Generator(device=dev).manual_seed(11)
    cases = []
    for n, t, u, a in shapes:
        x = torch.randn((n, t, u, a), generator=g, device=dev)
        lab = torch.randint(1, a, (n, u - 1), generator=g, device=dev, dtype=torch.int32)
        tl = torch.randint(t // 2, t + 1, (n,), generator=g, device=dev, dtype=torch.int32)
        ll = torch.randint((u - 1) // 2, u, (n,), generator=g, device=dev, dtype=torch.int32)
        costs, grads = torch.zeros(n), torch.zeros_like(x)
        assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, 0, 0) == 0
        cases.append((x, lab, tl, ll, costs.clone(), grads.clone()))
    torch.cuda.synchronize()
    errs = []
    start = threading.Barrier(len(cases))

    def worker(case):
        try:
            x, lab, tl, ll, want_c, want_g = case
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                start.wait()
                for _ in range(25):
                    costs, grads = torch.zeros(x.shape[0]), torch.empty_like(x)
                    assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, 0, 0) == 0
                    assert torch.equal(costs, want_c) and torch.equal(grads, want_g)
        except Exception as e:                                         # noqa: BLE001 -- reported to the main thread
            errs.append(repr(e))
    threads = [threading.Thread(target=worker, args=(c,)) for c in cases]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs


def test_determinism():
    acts, labels, tl, ll, blank = case_inputs("wide_u70")
    a = run_gpu(acts, labels, tl, ll, blank)
    b = run_gpu(acts, labels, tl, ll, blank)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ----------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties + oracle on a slice of the batch
FULL = {"c2": (16, 150, 41, 28, torch.float32), "c3": (128, 150, 21, 5000, torch.float32),
        "c4": (64, 1500, 301, 50, torch.float32), "c5": (128, 200, 41, 1024, torch.bfloat16)}


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_properties(oracle, name):
    from warprnnt_pytorch import warp_rnnt
    N, T, U, A, dtype = FULL[name]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand((N, T, U, A), generator=g, device=dev, dtype=torch.float32).to(dtype)
    labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
    tl = torch.randint(T // 2, T + 1, (N,), generator=g, device=dev, dtype=torch.int32)
    ll = torch.randint((U - 1) // 2, U, (N,), generator=g, device=dev, dtype=torch.int32)
    tl[0], ll[0] = T, U - 1
    costs = torch.zeros(N)
    grads = torch.full_like(x, 9.0)
    assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, costs, grads, 0, 0) == 0
    assert torch.isfinite(costs).all()
    # (1) every row of the dense logit gradient sums to zero (softmax-composed gradient)
    #     within the sum of the per-element quanta of that row (oracle.rowsum_bound; a row without its softmax term
    #     sums to about half of sum |g| and fails this by two orders of magnitude)
    gf = grads.float()
    over = gf.sum(-1).abs() / oracle.rowsum_bound(gf.abs().sum(-1), dtype)
    assert over.max().item() <= 1.0, over.max().item()
    if dtype == torch.float32:
        assert gf.sum(-1).abs().max().item() < 2e-4
    del gf, over
    # (2) padded region exactly zero
    t_idx = torch.arange(T, device=dev).view(1, T, 1)
    u_idx = torch.arange(U, device=dev).view(1, 1, U)
    pad = (t_idx >= tl.view(N, 1, 1)) | (u_idx > ll.view(N, 1, 1))
    assert grads.float().abs().amax(-1)[pad].max().item() == 0.0
    # (3) blank column of the first cell: -sum_v!=blank ... total outflow of (0,0) equals 1:
    #     g[0,0,blank] + g[0,0,y_0] - (softmax mass) telescopes to row-sum 0 (covered by (1)); instead
    #     check the occupancy identity  sum_u exp-occupancy on every anti-diagonal through the costs:
    #     forward-only scoring returns the same costs bit for bit.
    costs2 = torch.zeros(N)
    assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, costs2, torch.zeros(0, device=dev, dtype=dtype), 0, 0) == 0
    assert torch.equal(costs, costs2)
    # (4) the oracle (fp64, on the rounded inputs) on eight samples spread over this very batch
    import os
    oracle.lib().oracle_set_num_threads(min(64, os.cpu_count() or 8))
    pick = sorted(set(int(i) for i in np.linspace(0, N - 1, 8)))
    xs = x[pick].double().cpu().numpy()
    ref_c, ref_g, mag = oracle.rnnt_logits(xs, labels[pick].cpu().numpy(), tl[pick].cpu().numpy(), ll[pick].cpu().numpy(),
                                           want_mag=True)
    got_c = costs[pick].double().numpy()
    got_g = grads[pick].double().cpu().numpy()
    assert np.abs(got_c - ref_c).max() <= 1e-4 * np.abs(ref_c).max()               # loss: 1e-4 relative
    assert np.abs(got_g - ref_g).max() < (1e-3 if dtype == torch.float32 else 4e-3)  # grads: 1e-3 absolute (north_star) ...
    # ... which at A = 5000 / 1024 exceeds every non-blank / non-label entry: the check that can see them is per element
    r = oracle.assert_grads(got_g, ref_g, mag, dtype, what=name)
    # negative control: the same gradient without its softmax term passes the absolute bound above and must FAIL this one
    if name in ("c3", "c5"):
        bad = np.where(mag > np.abs(ref_g) * (1 + 1e-9), got_g, 0.0)                # keeps the blank / label columns only
        assert np.abs(bad - ref_g).max() < (1e-3 if dtype == torch.float32 else 4e-3)
        assert not oracle.grad_check(bad, ref_g, mag, dtype)["passed"]


# ----------------------------------------------------------------------------------------------
# structural boundaries of the kernels: wavefront edges (U = 63/64/65/128/129/200), chunk edges
# of the lattice sweep (T+U around multiples of 16), one-cell lattices, tile / row kernel switch
# (row bytes around 2 KB), packet straddling (A odd, A < packet), blank at either end
BOUNDARY_SHAPES = [
    (1, 1, 1, 1), (1, 1, 1, 7), (2, 1, 5, 9), (2, 6, 1, 9), (1, 2, 2, 2), (3, 16, 2, 3), (3, 17, 2, 4),
    (2, 15, 3, 31), (2, 31, 3, 33), (2, 33, 16, 8), (2, 5, 63, 6), (2, 5, 64, 6), (2, 5, 65, 6),
    (1, 40, 128, 5), (1, 40, 129, 5), (1, 9, 200, 4), (1, 300, 66, 3), (2, 48, 17, 511), (2, 12, 9, 512),
    (2, 12, 9, 513), (1, 7, 5, 2049), (3, 20, 33, 1), (2, 130, 70, 2),
]


@pytest.mark.parametrize("shape", BOUNDARY_SHAPES)
def test_boundary_shapes(oracle, shape):
    N, T, U, A = shape
    rng = np.random.default_rng(7 * N + 13 * T + 17 * U + 19 * A)
    acts = rng.standard_normal(shape) * 1.5
    blank = int(rng.integers(0, A))
    labels = rng.integers(0, A, size=(N, max(U - 1, 1)))[:, :U - 1]
    if A > 1:
        labels[labels == blank] = (blank + 1) % A
    tl = rng.integers(1, T + 1, size=N); tl[0] = T
    ll = rng.integers(0, U, size=N); ll[-1] = U - 1
    if N == 1:
        tl[0], ll[0] = T, U - 1
    if A == 1 and U > 1:          # only the blank exists: labels must equal it; the GPU contract
        labels[:] = 0             # subtracts both corrections, so compare the loss only
    labels = labels.reshape(N, U - 1) if U > 1 else np.zeros((N, 0), dtype=np.int64)
    a32 = acts.astype(np.float32).astype(np.float64)
    ref_c, ref_g = oracle.rnnt_logits(a32, labels if U > 1 else np.zeros((N, 1), dtype=np.int32)[:, :0],
                                      tl, ll, blank)
    lab_dev = labels if U > 1 else np.zeros((N, 0), dtype=np.int32)
    costs, grads = run_gpu(a32, lab_dev, tl, ll, blank)
    assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    if not (A == 1 and U > 1):
        assert np.abs(grads - ref_g).max() < 1e-4
    for b in range(N):
        assert not grads[b, tl[b]:].any() and not grads[b, :, ll[b] + 1:].any()
    c64, g64 = run_gpu(acts, lab_dev, tl, ll, blank, dtype=torch.float64)
    r64c, r64g = oracle.rnnt_logits(acts, labels if U > 1 else np.zeros((N, 1), dtype=np.int32)[:, :0], tl, ll, blank)
    assert np.abs(c64 - r64c).max() <= 1e-10 * max(1.0, np.abs(r64c).max())
    if not (A == 1 and U > 1):
        assert np.abs(g64 - r64g).max() < 1e-9


def test_random_sweep_fp32_and_bf16(oracle):
    """60 seeded random problems, variable lengths, fp32 and bf16 storage."""
    rng = np.random.default_rng(2026)
    for it in range(60):
        N = int(rng.integers(1, 5)); T = int(rng.integers(1, 70)); U = int(rng.integers(1, 140))
        A = int(rng.choice([2, 3, 5, 8, 17, 28, 50, 64, 100, 257, 1024]))
        acts = rng.standard_normal((N, T, U, A)) * float(rng.choice([0.5, 2.0, 6.0]))
        blank = int(rng.integers(0, A))
        labels = rng.integers(0, A, size=(N, U - 1))
        labels[labels == blank] = (blank + 1) % A
        tl = rng.integers(1, T + 1, size=N); tl[int(rng.integers(0, N))] = T
        ll = rng.integers(0, U, size=N); ll[int(rng.integers(0, N))] = U - 1
        dtype = torch.bfloat16 if it % 3 == 2 else torch.float32
        x = torch.tensor(acts).to(dtype)
        ref_c, ref_g, mag = oracle.rnnt_logits(x.double().numpy(), labels, tl, ll, blank, want_mag=True)
        costs, grads = run_gpu(x.double().numpy(), labels, tl, ll, blank, dtype=dtype)
        assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max()), (it, N, T, U, A)
        if dtype == torch.float32:
            assert np.abs(grads - ref_g).max() < 1e-4, (it, N, T, U, A, str(dtype))
        oracle.assert_grads(grads, ref_g, mag, dtype, what=(it, N, T, U, A, str(dtype)))


@pytest.mark.parametrize("shape", [(2, 3, 1024, 4),      # maxU at the limit: 16 wavefronts, 1024-thread blocks
                                   (1, 3000, 2, 6),      # very long utterance, 188 chunks
                                   (2, 3, 2, 100000),    # huge vocabulary: 400 KB rows
                                   (700, 3, 2, 5),       # many tiny samples
                                   (3, 40, 700, 3)])     # wide lattice, 11 wavefronts, tiny vocab (packet < row)
def test_extreme_shapes(oracle, shape):
    N, T, U, A = shape
    rng = np.random.default_rng(N + T + U + A)
    acts = (rng.standard_normal(shape) * 1.5).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1))
    tl = rng.integers(max(1, T // 2), T + 1, size=N); tl[0] = T
    ll = rng.integers((U - 1) // 2, U, size=N); ll[-1] = U - 1
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll)
    costs, grads = run_gpu(acts, labels, tl, ll)
    assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    assert np.abs(grads - ref_g).max() < 2e-4
    assert np.abs(grads.sum(-1)).max() < 2e-4          # every row of the logit gradient sums to zero


def test_limits_are_reported_not_crashed():
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    x = torch.zeros(8, device=dev)
    i = torch.ones(4, dtype=torch.int32, device=dev)
    costs = torch.zeros(1)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, maxT=1, maxU=1025, batch_first=True)
    st = lib.compute_rnnt_loss(x.data_ptr(), None, i.data_ptr(), i.data_ptr(), i.data_ptr(), 2, 1, costs.data_ptr(),
                               x.data_ptr(), opt)
    assert st == 2            # maxU > 1024 -> RNNT_STATUS_INVALID_VALUE (documented limit, as the reference)


def test_async_entry_is_graph_capturable(oracle):
    """compute_rnnt_loss_async only enqueues on the given stream (no allocation, no sync, no host
    copy), so it can be captured in a HIP graph and replayed on new inputs."""
    from warprnnt_pytorch import _lib, warp_rnnt
    acts, labels, tl, ll, blank = case_inputs("var_a40")
    dev = torch.device("cuda:0")
    x = torch.tensor(acts, dtype=torch.float32, device=dev)
    lab, tlen, llen = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
    costs = torch.zeros(x.shape[0], device=dev)
    grads = torch.zeros_like(x)
    N, T, U, A = x.shape
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside capture
        warp_rnnt.gpu_rnnt_async(x, lab, tlen, llen, costs, grads, blank, workspace=ws)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        warp_rnnt.gpu_rnnt_async(x, lab, tlen, llen, costs, grads, blank, workspace=ws)
    for scale in (1.0, 0.5):                            # replay on changed inputs
        x.copy_(torch.tensor(acts * scale, dtype=torch.float32))
        costs.zero_(); grads.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ref_c, ref_g = oracle.rnnt_logits((acts * scale).astype(np.float32).astype(np.float64), labels, tl, ll, blank)
        assert np.abs(costs.cpu().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
        assert np.abs(grads.cpu().numpy() - ref_g).max() < 1e-4


def _fastemit_expectation(oracle, acts, labels, tl, ll, blank, lam):
    """FastEmit through the oracle: the sparse log-prob gradient's LABEL entries times (1 + lambda), then
    the chain rule through log_softmax (include/rnnt.h, compute_rnnt_loss_fastemit)."""
    lp = oracle.log_softmax(acts.astype(np.float64))
    costs, g_lp = oracle.rnnt_logprobs(lp, labels, tl, ll, blank)
    for b in range(acts.shape[0]):
        for u in range(ll[b]):
            g_lp[b, :tl[b], u, labels[b, u]] *= 1.0 + lam
    return costs, oracle.chain_rule_to_logits(lp, g_lp)


@pytest.mark.parametrize("shape,lam", [((3, 12, 6, 20), 0.01), ((2, 40, 70, 33), 0.5), ((2, 9, 5, 5000), 0.05)])
def test_fastemit_entry(oracle, shape, lam):
    from warprnnt_pytorch import warp_rnnt
    N, T, U, A = shape
    rng = np.random.default_rng(A + T)
    acts = rng.standard_normal(shape).astype(np.float32)
    blank = A - 1
    labels = rng.integers(0, A - 1, size=(N, U - 1)).astype(np.int32)
    tl = np.array([T] + list(rng.integers(max(1, T // 2), T + 1, size=N - 1)), dtype=np.int32)
    ll = np.array(list(rng.integers(U // 2, U, size=N - 1)) + [U - 1], dtype=np.int32)
    ref_c, ref_g = _fastemit_expectation(oracle, acts, labels, tl, ll, blank, lam)
    dev = torch.device("cuda:0")
    x = torch.tensor(acts, device=dev)
    args = [torch.tensor(a, device=dev) for a in (labels, tl, ll)]
    costs = torch.empty(N, device=dev)
    grads = torch.empty_like(x)
    ws = warp_rnnt.gpu_rnnt_async(x, args[0], args[1], args[2], costs, grads, blank, fastemit_lambda=lam)
    torch.cuda.synchronize()
    assert np.abs(costs.cpu().numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()      # the plain likelihood
    assert np.abs(grads.cpu().numpy() - ref_g).max() < 2e-4
    # lambda = 0 through the same entry is the plain loss, bit for bit
    g0, g1, c1 = torch.empty_like(x), torch.empty_like(x), torch.empty(N, device=dev)
    warp_rnnt.gpu_rnnt_async(x, args[0], args[1], args[2], c1, g0, blank, workspace=ws)
    lib = __import__("warprnnt_pytorch")._lib
    opt = lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=blank,
                          maxT=T, maxU=U, batch_first=True)
    call = lambda lam_: lib.lib().compute_rnnt_loss_fastemit(
        x.data_ptr(), g1.data_ptr(), args[0].data_ptr(), args[2].data_ptr(), args[1].data_ptr(), A, N, c1.data_ptr(),
        None, ws.data_ptr(), opt, 0, lam_)
    assert call(0.0) == 0
    torch.cuda.synchronize()
    assert torch.equal(g0, g1)
    assert call(-0.5) == 2 and call(float("nan")) == 2


def test_fastemit_in_the_module(oracle):
    from warprnnt_pytorch import RNNTLoss
    shape, lam = (4, 15, 7, 50), 0.02
    N, T, U, A = shape
    rng = np.random.default_rng(5)
    acts = rng.standard_normal(shape).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl = np.full(N, T, dtype=np.int32); tl[1] = T - 4
    ll = np.full(N, U - 1, dtype=np.int32); ll[2] = 2
    ref_c, ref_g = _fastemit_expectation(oracle, acts, labels, tl, ll, 0, lam)
    dev = torch.device("cuda:0")
    x = torch.tensor(acts, device=dev, requires_grad=True)
    loss = RNNTLoss(blank=0, reduction="mean", fastemit_lambda=lam)(
        x, *(torch.tensor(a, device=dev) for a in (labels, tl, ll)))
    loss.backward()
    assert abs(loss.item() - ref_c.mean()) <= 1e-4 * abs(ref_c.mean())
    assert np.abs(x.grad.cpu().numpy() - ref_g / N).max() < 1e-4
    with pytest.raises(NotImplementedError):            # CPU location: not offered
        RNNTLoss(fastemit_lambda=lam)(torch.tensor(acts), *(torch.tensor(a) for a in (labels, tl, ll)))


def test_minus_inf_logits(oracle):
    """Masked vocabulary entries (-inf logits) that are neither the blank nor a label: probability
    zero, gradient exactly zero there, everything else as the oracle says."""
    rng = np.random.default_rng(99)
    N, T, U, A = 2, 9, 4, 40
    acts = rng.standard_normal((N, T, U, A)).astype(np.float32)
    labels = rng.integers(1, 10, size=(N, U - 1))
    acts[..., 20:30] = -np.inf
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, [T, T], [U - 1, U - 1])
    costs, grads = run_gpu(acts, labels, [T, T], [U - 1, U - 1])
    assert np.isfinite(costs).all() and np.abs(costs - ref_c).max() < 1e-4 * np.abs(ref_c).max()
    assert not np.isnan(grads).any() and not grads[..., 20:30].any()
    assert np.abs(grads - ref_g).max() < 1e-4


def test_sharded_wrapper_over_rccl_single_rank():
    """ShardedRNNTLoss on the GPU with backend "nccl" (= RCCL on ROCm), world_size 1: async entry,
    device-resident costs, the all-reduce of [sum, count] on the compute stream.  (Two ranks are
    covered on CPU by tests/test_sharded_gloo.py; a single box has one GPU.)"""
    import os
    import torch.distributed as dist
    from warprnnt_pytorch import RNNTLoss
    from warprnnt_pytorch.sharded import ShardedRNNTLoss
    acts, labels, tl, ll, blank = case_inputs("blank5_a19")
    tl[:] = acts.shape[1]; ll[:] = acts.shape[2] - 1
    dev = torch.device("cuda:0")
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        args = [torch.tensor(a, device=dev) for a in (labels, tl, ll)]
        for reduction in ("mean", "sum", "none"):
            x1 = torch.tensor(acts, dtype=torch.float32, device=dev, requires_grad=True)
            x2 = x1.detach().clone().requires_grad_(True)
            l1 = ShardedRNNTLoss(blank=blank, reduction=reduction)(x1, *args)
            l2 = RNNTLoss(blank=blank, reduction=reduction)(x2, *args)
            l1.sum().backward(); l2.sum().backward()
            assert l1.is_cuda and torch.allclose(l1, l2, rtol=1e-6, atol=1e-5)
            assert torch.allclose(x1.grad, x2.grad, rtol=1e-5, atol=1e-6)
    finally:
        if created:
            dist.destroy_process_group()


def test_label_equal_to_blank_is_the_true_derivative(oracle):
    """A label that EQUALS the blank symbol: both corrections of gpu_rnnt_kernel.h:165-173 hit the same logit
    (the reference's CPU backward overwrites one with the other instead, cpu_rnnt.h:256-263, so its gradients
    are not a derivative there).  The forward cost is unambiguous, so the check is a central finite difference
    of the oracle's fp64 COST against this library's fp64 gradients -- every logit of a small problem."""
    rng = np.random.default_rng(12)
    N, T, U, A, blank = 2, 4, 3, 4, 2
    acts = rng.standard_normal((N, T, U, A))
    labels = np.array([[blank, 1], [3, blank]], dtype=np.int32)
    tl = np.array([4, 3], dtype=np.int32)
    ll = np.array([2, 2], dtype=np.int32)
    costs, grads = run_gpu(acts, labels, tl, ll, blank, dtype=torch.float64)
    ref_c, _ = oracle.rnnt_logits(acts, labels, tl, ll, blank, want_grad=False)
    assert np.allclose(costs, ref_c, rtol=1e-10)
    eps = 1e-5
    num = np.zeros_like(acts)
    for idx in np.ndindex(*acts.shape):
        hi, lo = acts.copy(), acts.copy()
        hi[idx] += eps
        lo[idx] -= eps
        ch, _ = oracle.rnnt_logits(hi, labels, tl, ll, blank, want_grad=False)
        cl, _ = oracle.rnnt_logits(lo, labels, tl, ll, blank, want_grad=False)
        num[idx] = (ch[idx[0]] - cl[idx[0]]) / (2 * eps)
    assert np.abs(grads - num).max() < 1e-8
    c32, g32 = run_gpu(acts.astype(np.float32), labels, tl, ll, blank)
    assert np.abs(g32 - num).max() < 1e-5



def test_stage_ranges_for_external_profilers(tmp_path):
    """rnnt_profile_enable(2) / WARPRNNT_ROCTX=1: every call brackets the enqueue of its stages with roctx ranges (the
    counterpart of the reference's DEBUG_TIME stage timers, include/detail/gpu_rnnt.h:112-122).  Results are the same
    with the ranges on; under `rocprofv3 --marker-trace` the four stage names appear in the marker trace."""
    import os
    import shutil
    import subprocess
    import sys
    from warprnnt_pytorch import _lib
    acts, labels, tl, ll, blank = case_inputs("var_a40")
    base_c, base_g = run_gpu(acts, labels, tl, ll, blank)
    lib = _lib.lib()
    lib.rnnt_profile_enable(2)
    try:
        c, g = run_gpu(acts, labels, tl, ll, blank)
    finally:
        lib.rnnt_profile_enable(0)
    assert np.array_equal(c, base_c) and np.array_equal(g, base_g)
    prof = shutil.which("rocprofv3")
    if prof is None:
        pytest.skip("rocprofv3 not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np, torch\n"
            "from tests.test_gpu_parity import run_gpu, case_inputs\n"
            "a, l, t, u, b = case_inputs('var_a40')\n"
            "run_gpu(a, l, t, u, b)\n" % (root, os.path.join(root, "warp-transducer_amd")))
    env = dict(os.environ, WARPRNNT_ROCTX="1", TMPDIR=str(tmp_path))
    try:
        out = subprocess.run([prof, "--marker-trace", "--kernel-trace", "--output-format", "csv", "-d", str(tmp_path / "prof"), "--",
                              sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    except (OSError, subprocess.TimeoutExpired) as e:
        pytest.skip("rocprofv3 could not be run here: %r" % (e,))
    text, traces = "", 0
    for dirpath, _, files in os.walk(str(tmp_path / "prof")):
        for fn in files:
            if fn.endswith(".csv"):
                traces += 1
                if "marker" in fn:
                    text += open(os.path.join(dirpath, fn)).read()
    if out.returncode != 0 or traces == 0:                 # the profiler itself did not work in this environment
        pytest.skip("rocprofv3 failed here (rc %d): %s" % (out.returncode, out.stderr[-300:]))
    for name in ("warprnnt:row_stats", "warprnnt:lattice", "warprnnt:coefficients", "warprnnt:gradient"):
        assert name in text, (name, text[:500])


@pytest.mark.parametrize("dtype,A", [(torch.bfloat16, 1024), (torch.float32, 1000), (torch.float16, 2048)])
def test_padding_flag_follows_the_batch(oracle, dtype, A):
    """Rows of 2-8 KB: the gradient kernel leaves the logits of padded rows unread only when the coefficient kernel has seen
    padding in THIS batch (one flag word in the workspace, zeroed by the lattice kernel of every call).  Ragged, full-length
    and ragged again on one workspace, with NaNs planted in the padded rows of the ragged batches (never read, never
    propagated), each against the oracle; then the per-sample scale form and the two-phase entries on the same workspace."""
    from warprnnt_pytorch import warp_rnnt, _lib
    dev = torch.device("cuda:0")
    N, T, U = 5, 14, 9
    rng = np.random.default_rng(A)
    base = rng.standard_normal((N, T, U, A)).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    ragged = (np.array([T, T - 3, T // 2, T - 1, 5], dtype=np.int32), np.array([U - 1, 3, U - 1, 0, U - 2], dtype=np.int32))
    full = (np.full(N, T, dtype=np.int32), np.full(N, U - 1, dtype=np.int32))
    esz = 4 if dtype == torch.float32 else 2
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, esz), dtype=torch.uint8, device=dev)
    lab = torch.tensor(labels, device=dev)

    def inputs(lens, poison):
        x = torch.tensor(base, device=dev).to(dtype)
        if poison:
            for b in range(N):
                x[b, lens[0][b]:] = float("nan")
                x[b, :, lens[1][b] + 1:] = float("nan")
        clean = torch.tensor(base, device=dev).to(dtype).float().cpu().numpy().astype(np.float64)
        return x, clean

    for lens, poison in ((ragged, True), (full, False), (ragged, True)):
        x, clean = inputs(lens, poison)
        ref_c, ref_g, mag = oracle.rnnt_logits(clean, labels, lens[0], lens[1], 0, want_mag=True)
        costs, grads = torch.zeros(N), torch.full_like(x, 7.0)
        tl, ll = torch.tensor(lens[0], device=dev), torch.tensor(lens[1], device=dev)
        assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, 0, 0, workspace=ws) == 0
        g = grads.float().cpu().numpy().astype(np.float64)
        assert np.abs(costs.numpy() - ref_c).max() <= 1e-4 * np.abs(ref_c).max()
        oracle.assert_grads(g, ref_g, mag, dtype)              # per element; NaN fails it
        for b in range(N):
            assert not g[b, lens[0][b]:].any() and not g[b, :, lens[1][b] + 1:].any()
    # per-sample scale folded in (the form autograd callers get), ragged with poisoned padding, same workspace
    x, clean = inputs(ragged, True)
    ref_c, ref_g, mag = oracle.rnnt_logits(clean, labels, ragged[0], ragged[1], 0, want_mag=True)
    tl, ll = torch.tensor(ragged[0], device=dev), torch.tensor(ragged[1], device=dev)
    scale = torch.tensor([0.5, 2.0, 1.0, 0.25, 3.0], device=dev)
    dcosts, grads = torch.zeros(N, device=dev), torch.full_like(x, 7.0)
    warp_rnnt.gpu_rnnt_async(x, lab, tl, ll, dcosts, grads, 0, grad_scale=scale, workspace=ws)
    torch.cuda.synchronize()
    g = grads.float().cpu().numpy().astype(np.float64)
    sc = scale.cpu().numpy().reshape(N, 1, 1, 1).astype(np.float64)
    oracle.assert_grads(g, ref_g * sc, mag * sc, dtype, scale=3.0)
    # two-phase entries: the flag written in the forward phase is what the backward phase reads
    dcosts2, grads2 = torch.zeros(N, device=dev), torch.full_like(x, 7.0)
    ws2 = warp_rnnt.gpu_rnnt_fwd(x, lab, tl, ll, dcosts2, 0, True)
    warp_rnnt.gpu_rnnt_bwd(x, grads2, scale, ws2, 0)
    torch.cuda.synchronize()
    assert torch.equal(dcosts, dcosts2) and torch.equal(grads, grads2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("shape", [(3, 17, 6, 40),        # flat packets, several rows per packet chunk
                                   (2, 9, 5, 5003),       # long unaligned rows: straddling packets, row-skipping form
                                   (4, 12, 7, 1024),      # 2-4 KB rows: the padflag form
                                   (2, 33, 70, 12),       # tiny vocabulary, 2-D statistics tiles
                                   (2, 810, 4, 16)])      # long lattice (two-half schedule when an aux stream is set)
def test_gradients_in_place(oracle, shape, dtype):
    """`gradients == activations` (rnnt.h, IN PLACE): the gradient overwrites the logits, bit-identical to the out-of-place
    call -- one-call entry, scaled asynchronous entry, two-phase pair, packed layout, and the row-form kernel of an
    unaligned tensor; tensors that overlap partially are refused.  The reference memsets the gradient tensor before it
    reads the activations (include/detail/gpu_rnnt.h:107-110) and cannot do this."""
    from warprnnt_pytorch import warp_rnnt, _lib
    from warprnnt_pytorch.packed import pack_joint, row_offsets
    dev = torch.device("cuda:0")
    N, T, U, A = shape
    rng = np.random.default_rng(sum(shape))
    base = torch.tensor(rng.standard_normal(shape) * 2.0, device=dev).to(dtype)
    labels = torch.tensor(rng.integers(1, A, size=(N, U - 1)).astype(np.int32), device=dev)
    tl_np = rng.integers(max(1, T // 2), T + 1, size=N).astype(np.int32); tl_np[0] = T
    ll_np = rng.integers(0, U, size=N).astype(np.int32); ll_np[-1] = U - 1
    tl, ll = torch.tensor(tl_np, device=dev), torch.tensor(ll_np, device=dev)
    cdt = torch.float64 if dtype == torch.float64 else torch.float32
    # one-call entry
    c0, g0 = torch.zeros(N, dtype=cdt), torch.full_like(base, 3.0)
    assert warp_rnnt.gpu_rnnt(base, labels, tl, ll, c0, g0, 0, 0) == 0
    x = base.clone()
    c1 = torch.zeros(N, dtype=cdt)
    assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, c1, x, 0, 0) == 0
    assert torch.equal(c0, c1) and torch.equal(g0.view(torch.uint8), x.view(torch.uint8))
    ref_c, ref_g, mag = oracle.rnnt_logits(base.double().cpu().numpy(), labels.cpu().numpy(), tl_np, ll_np, want_mag=True)
    oracle.assert_grads(x.double().cpu().numpy(), ref_g, mag, dtype, rel=1e-3 if dtype != torch.float64 else None)
    # asynchronous entry with a per-sample scale folded in
    scale = torch.tensor(rng.uniform(0.25, 2.0, size=N), device=dev, dtype=cdt)
    dc0, gs0 = torch.zeros(N, device=dev, dtype=cdt), torch.empty_like(base)
    warp_rnnt.gpu_rnnt_async(base, labels, tl, ll, dc0, gs0, 0, grad_scale=scale)
    x = base.clone(); dc1 = torch.zeros(N, device=dev, dtype=cdt)
    warp_rnnt.gpu_rnnt_async(x, labels, tl, ll, dc1, x, 0, grad_scale=scale)
    torch.cuda.synchronize()
    assert torch.equal(dc0, dc1) and torch.equal(gs0.view(torch.uint8), x.view(torch.uint8))
    # two-phase pair: the forward phase leaves only the workspace, the backward phase overwrites the logits
    x = base.clone(); dc2 = torch.zeros(N, device=dev, dtype=cdt)
    ws = warp_rnnt.gpu_rnnt_fwd(x, labels, tl, ll, dc2, 0, True)
    warp_rnnt.gpu_rnnt_bwd(x, x, scale, ws, 0)
    torch.cuda.synchronize()
    assert torch.equal(dc0, dc2) and torch.equal(gs0.view(torch.uint8), x.view(torch.uint8))
    # packed layout
    if dtype != torch.float64:
        p0 = pack_joint(base, tl, ll).contiguous()
        offs = row_offsets(tl, ll)
        esz = 4 if dtype == torch.float32 else 2
        code = _lib.DT_F32 if dtype == torch.float32 else _lib.DT_BF16
        wsp = torch.empty(_lib.workspace_bytes(T, U, N, True, esz), dtype=torch.uint8, device=dev)
        opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=0,
                               maxT=T, maxU=U, batch_first=True)
        gp0, cp0 = torch.empty_like(p0), torch.zeros(N, device=dev)
        call = lambda a, g, c: _lib.lib().compute_rnnt_loss_packed(a.data_ptr(), g.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(),
                                                                   offs.data_ptr(), a.shape[0], A, N, c.data_ptr(), None, wsp.data_ptr(), opt, code, 0.0)
        assert call(p0, gp0, cp0) == 0
        p1, cp1 = p0.clone(), torch.zeros(N, device=dev)
        assert call(p1, p1, cp1) == 0
        torch.cuda.synchronize()
        assert torch.equal(cp0, cp1) and torch.equal(gp0.view(torch.uint8), p1.view(torch.uint8))
    # an unaligned tensor (base pointer one element off a 16-byte boundary): the row-form gradient kernel
    buf = torch.zeros(base.numel() + 1, device=dev, dtype=dtype)
    xu = buf[1:].view(shape); xu.copy_(base)
    cu = torch.zeros(N, dtype=cdt)
    assert warp_rnnt.gpu_rnnt(xu, labels, tl, ll, cu, xu, 0, 0) == 0
    assert torch.allclose(c0, cu, rtol=1e-6, atol=0)       # (other statistics forms for an unaligned tensor: last-bit differences)
    oracle.assert_grads(xu.double().cpu().numpy(), ref_g, mag, dtype, rel=1e-3 if dtype != torch.float64 else None)
    # partial overlap: refused, nothing launched
    big = torch.zeros(base.numel() + 64, device=dev, dtype=dtype)
    xa, ga = big[:base.numel()].view(shape), big[64:].view(shape)
    xa.copy_(base)
    st = _lib.lib().compute_rnnt_loss_async(xa.data_ptr(), ga.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N,
                                            dc1.data_ptr(), None, ws.data_ptr(),
                                            _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream,
                                                             blank_label=0, maxT=T, maxU=U, batch_first=True),
                                            {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float64: _lib.DT_F64}[dtype])
    assert st == _lib.RNNT_STATUS_INVALID_VALUE
    torch.cuda.synchronize()
    assert torch.equal(xa, base)


@pytest.mark.parametrize("loader", ["ext", "ctypes"])
def test_binding_rejects_host_side_arguments_and_short_workspaces(monkeypatch, loader):
    """labels / lengths on the host next to device activations used to be a GPU fault, and a caller-owned
    workspace was passed on unchecked (its layout is private and grew by the per-sample `poison` array).  Both are
    exceptions now, in the compiled module and in the ctypes loader, in the one-call and the two-phase entries and in the
    additive joint."""
    from warprnnt_pytorch import RNNTLoss, _lib, warp_rnnt
    from warprnnt_pytorch.add_network import RNNTLossAdd
    if loader == "ctypes":
        monkeypatch.setattr(warp_rnnt, "_EXT", None)
    dev = torch.device("cuda:0")
    N, T, U, A = 2, 5, 3, 8
    x = torch.randn(N, T, U, A, device=dev)
    lab = torch.ones(N, U - 1, dtype=torch.int32, device=dev)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    for bad in ((lab.cpu(), tl, ll), (lab, tl.cpu(), ll), (lab, tl, ll.cpu())):
        with pytest.raises(ValueError, match="must be on the device"):
            RNNTLoss()(x, *bad)
        with pytest.raises(ValueError, match="must be on the device"):
            warp_rnnt.gpu_rnnt(x, *[bad[0], bad[1], bad[2]], torch.zeros(N), torch.empty_like(x), 0, 0)
        with pytest.raises(ValueError, match="must be on the device"):
            RNNTLossAdd()(torch.randn(N, T, A, device=dev), torch.randn(N, U, A, device=dev), *bad)
    need = _lib.workspace_bytes(T, U, N, True, 4)
    costs, grads = torch.zeros(N), torch.empty_like(x)
    with pytest.raises(ValueError, match="workspace"):
        warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, 0, 0, workspace=torch.empty(need - 1, dtype=torch.uint8, device=dev))
    with pytest.raises(ValueError, match="workspace"):
        warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, 0, 0, workspace=torch.empty(need, dtype=torch.uint8))
    ws = warp_rnnt.gpu_rnnt_fwd(x, lab, tl, ll, torch.zeros(N, device=dev), 0, True)
    with pytest.raises(ValueError, match="workspace"):
        warp_rnnt.gpu_rnnt_bwd(x, grads, None, ws[:need // 2], 0)
    assert warp_rnnt.gpu_rnnt(x, lab, tl, ll, costs, grads, 0, 0, workspace=torch.empty(need, dtype=torch.uint8, device=dev)) == 0
    torch.cuda.synchronize()
