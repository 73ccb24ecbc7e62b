"""warprnnt_pytorch on CPU tensors -- pytorch_binding/test/test.py (small_test :52-78, big_test
:80-161) restated, plus reductions, validation errors and the extension-module surface."""
import numpy as np
import pytest
import torch

from tests.golden import literals as G
from warprnnt_pytorch import RNNTLoss, certify_inputs, rnnt_loss, warp_rnnt


def wrap_and_call(fn, acts, labels, dtype=torch.float32):
    acts = torch.tensor(acts, dtype=dtype, requires_grad=True)
    lengths = torch.IntTensor([acts.shape[1]] * acts.shape[0])
    label_lengths = torch.IntTensor([len(l) for l in labels])
    costs = fn(acts, torch.IntTensor(labels), lengths, label_lengths)
    torch.sum(costs).backward()
    return costs.data.numpy(), acts.grad.data.numpy()


def test_small_test():
    cost, grads = wrap_and_call(RNNTLoss(reduction='sum'), G.SMALL_ACTS, [[1, 2]])
    assert np.allclose(cost, G.SMALL_COST, rtol=1e-6)
    assert np.allclose(grads, G.SMALL_GRADS)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_big_test(dtype):
    costs, grads = wrap_and_call(RNNTLoss(reduction='sum'), G.BIG_ACTS, [[1, 2], [1, 1]], dtype)
    assert np.allclose(costs, sum(G.OPTIONS_COSTS))
    assert np.allclose(grads, G.BIG_GRADS, rtol=1e-3)


def test_reductions_and_functional_form():
    labels = [[1, 2], [1, 1]]
    none, g_none = wrap_and_call(RNNTLoss(reduction='none'), G.BIG_ACTS, labels)
    assert none.shape == (2,) and np.allclose(none, G.OPTIONS_COSTS, atol=1e-5)
    s, g_sum = wrap_and_call(RNNTLoss(reduction='sum'), G.BIG_ACTS, labels)
    m, g_mean = wrap_and_call(lambda *a: rnnt_loss(*a), G.BIG_ACTS, labels)      # default reduction='mean'
    assert s.shape == (1,) and m.shape == (1,)
    assert np.allclose(s, none.sum()) and np.allclose(m, none.sum() / 2)
    assert np.allclose(g_sum, g_none) and np.allclose(g_mean, g_sum / 2)


def test_no_grad_scoring_path():
    acts = torch.tensor(G.SMALL_ACTS, dtype=torch.float32)
    out = RNNTLoss(reduction='none')(acts, torch.IntTensor([[1, 2]]), torch.IntTensor([2]), torch.IntTensor([2]))
    assert abs(out.item() - G.SMALL_COST) < 1e-4


def test_certify_inputs_errors():
    acts = torch.zeros(2, 4, 3, 3)
    lab, tl, ll = torch.IntTensor([[1, 2], [1, 1]]), torch.IntTensor([4, 4]), torch.IntTensor([2, 2])
    certify_inputs(acts, lab, tl, ll)
    with pytest.raises(TypeError):
        certify_inputs(acts, lab.long(), tl, ll)
    with pytest.raises(TypeError):
        certify_inputs(acts, lab, tl.long(), ll)
    with pytest.raises(ValueError):
        certify_inputs(acts.transpose(1, 2), lab, tl, ll)                 # not contiguous
    with pytest.raises(ValueError):
        certify_inputs(acts[0], lab, tl[:1], ll[:1])                      # wrong rank / batch
    with pytest.raises(ValueError):
        certify_inputs(acts, lab, torch.IntTensor([3, 3]), ll)            # T != max(act_lens)
    with pytest.raises(ValueError):
        certify_inputs(acts, lab, tl, torch.IntTensor([1, 1]))            # U != max(label_lens)+1
    with pytest.raises(ValueError):
        certify_inputs(acts, lab, tl[:1], ll)


def test_extension_module_surface():
    # binding.cpp:12-19,157-162: cpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, blank, threads)
    lp = torch.log_softmax(torch.tensor(G.OPTIONS_ACTS_6DP, dtype=torch.float32), -1)
    costs, grads = torch.zeros(2), torch.zeros_like(lp)
    rc = warp_rnnt.cpu_rnnt(lp, torch.IntTensor([[1, 2], [1, 1]]), torch.IntTensor([4, 4]), torch.IntTensor([2, 2]),
                            costs, grads, 0, 1)
    assert rc == 0
    assert np.allclose(costs.numpy(), G.OPTIONS_COSTS, atol=1e-4)
    assert np.abs(grads.numpy() - G.OPTIONS_LOGPROB_GRADS).max() < 1e-4
    assert warp_rnnt.cpu_rnnt(lp.half(), None, None, None, None, None, 0, 1) == -1   # unsupported dtype
    assert hasattr(warp_rnnt, "gpu_rnnt")


def test_compiled_module_exports_and_additive_joint_checks():
    """The compiled extension module carries the reference's two functions, the two-phase pair and both losses as C++ autograd
    functions; the additive-joint loss refuses host tensors and wrong dtypes with the same exception types from the Python
    checks (the C++ twin of these checks runs on the GPU box: tests/test_gpu_add_network.py::test_both_bindings)."""
    import pytest
    ext = getattr(warp_rnnt, "_EXT", None)
    assert ext is not None, "the compiled extension module was not built (python warp-transducer_amd/warprnnt_pytorch/build_ext.py)"
    for name in ("cpu_rnnt", "gpu_rnnt", "gpu_rnnt_fwd", "gpu_rnnt_bwd", "certify_inputs", "rnnt_loss", "rnnt_loss_add", "library_version"):
        assert hasattr(ext, name), name
    assert ext.library_version() == 1
    from warprnnt_pytorch.add_network import RNNTLossAdd
    f, g = torch.zeros(1, 4, 5), torch.zeros(1, 3, 5)
    lab, tl, ll = torch.IntTensor([[1, 2]]), torch.IntTensor([4]), torch.IntTensor([2])
    with pytest.raises(ValueError, match="GPU only"):
        RNNTLossAdd()(f, g, lab, tl, ll)
    with pytest.raises(ValueError, match="GPU only"):
        RNNTLossAdd(validate=False)(f, g, lab, tl, ll)
    with pytest.raises(TypeError, match="labels must be"):
        RNNTLossAdd()(f, g, lab.long(), tl, ll)
    with pytest.raises(ValueError, match="must be 3D"):
        RNNTLossAdd()(f[0], g, lab, tl, ll)
