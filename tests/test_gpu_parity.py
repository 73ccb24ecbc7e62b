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
