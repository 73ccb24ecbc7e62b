"""-m gpu: the linear-domain lattice kernel (one-wavefront fp32 lattices, csrc/rnnt_kernels.h lattice_lin_kernel) against
the CPU oracle -- long lattices (many power-of-two re-normalisations), lengths either side of its 12-diagonal chunks, and
inputs that leave the range its fp64 probabilities cover, which must come out of the log-domain fallback unchanged."""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import run_gpu

pytestmark = pytest.mark.gpu


def _case(rng, N, T, U, A, scale, ragged=True):
    acts = rng.standard_normal((N, T, U, A)) * scale
    labels = rng.integers(1, A, size=(N, U - 1))
    act_lens = rng.integers(max(1, T // 2), T + 1, size=N) if ragged else np.full(N, T)
    label_lens = rng.integers(0, U, size=N) if ragged else np.full(N, U - 1)
    act_lens[0], label_lens[-1] = T, U - 1
    return acts.astype(np.float32), labels, act_lens, label_lens


def _check(oracle, acts, labels, act_lens, label_lens, tol_c=1e-4, tol_g=1e-4):
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, act_lens, label_lens, 0)
    costs, grads = run_gpu(acts, labels, act_lens, label_lens, 0, torch.float32)
    assert np.isfinite(costs).all()
    assert np.abs(costs - ref_c).max() <= tol_c * max(1.0, np.abs(ref_c).max())
    assert np.abs(grads - ref_g).max() <= tol_g
    return costs


@pytest.mark.parametrize("shape", [(3, 700, 40, 6), (2, 1500, 12, 5), (4, 300, 64, 4)])
def test_long_lattices(oracle, shape):
    """Hundreds of chunks: the accumulated exponent reaches thousands of bits, the likelihood must not drift."""
    rng = np.random.default_rng(shape[1])
    _check(oracle, *_case(rng, *shape, scale=1.5))


@pytest.mark.parametrize("T", [1, 2, 11, 12, 13, 23, 24, 25, 36, 37])
@pytest.mark.parametrize("U", [1, 2, 13, 64])
def test_lengths_around_the_chunk_size(oracle, T, U):
    rng = np.random.default_rng(100 * T + U)
    _check(oracle, *_case(rng, 3, T, U, 7, scale=2.0))


def test_every_sample_length_combination_in_one_batch(oracle):
    """T_b and U_b swept inside one padded batch: masked cells, finished columns, one-cell lattices."""
    T, U, A = 14, 14, 5
    rng = np.random.default_rng(5)
    pairs = [(t, u) for t in range(1, T + 1, 2) for u in range(0, U, 3)] + [(T, U - 1)]
    N = len(pairs)
    acts = (rng.standard_normal((N, T, U, A)) * 2).astype(np.float32)
    labels = rng.integers(1, A, size=(N, U - 1))
    _check(oracle, acts, labels, np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs]))


@pytest.mark.parametrize("scale", [60.0, 300.0])
def test_log_probs_below_the_fp32_exponent_range(oracle, scale):
    """log-softmax values far below -126 bits: the operand-side guard must hand the sample to the log-domain sweep."""
    rng = np.random.default_rng(int(scale))
    acts, labels, tl, ll = _case(rng, 4, 40, 20, 9, scale=scale)
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll, 0)
    costs, grads = run_gpu(acts, labels, tl, ll, 0, torch.float32)
    assert np.isfinite(costs).all()
    assert np.abs(costs - ref_c).max() <= 1e-5 * np.abs(ref_c).max()      # (costs of tens of thousands: relative)
    assert np.abs(grads - ref_g).max() <= 2e-3      # (an fp32 lattice holding values of several thousand: ulp 2.4e-4)


def test_underflow_inside_a_chunk(oracle):
    """Every step costs ~100 bits (all log-probs representable on their own): twelve steps leave fp64's range between two
    re-normalisations, the result-side guard must catch it."""
    rng = np.random.default_rng(8)
    N, T, U, A = 2, 60, 8, 4
    acts = np.zeros((N, T, U, A), dtype=np.float32)
    acts[..., 3] = 70.0                                     # blank and every label sit ~101 bits below the winner
    acts += rng.standard_normal(acts.shape).astype(np.float32) * 0.5
    labels = rng.integers(1, 3, size=(N, U - 1))
    tl, ll = np.array([T, T - 7]), np.array([U - 1, U - 3])
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll, 0)
    costs, grads = run_gpu(acts, labels, tl, ll, 0, torch.float32)
    assert np.abs(costs - ref_c).max() <= 1e-5 * np.abs(ref_c).max()
    assert np.abs(grads - ref_g).max() <= 2e-3


def test_mixed_batch_only_some_samples_fall_back(oracle):
    rng = np.random.default_rng(9)
    acts, labels, tl, ll = _case(rng, 6, 50, 30, 11, scale=2.0)
    acts[1] *= 100.0
    acts[4, 10:20] *= 200.0
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll, 0)
    costs, grads = run_gpu(acts, labels, tl, ll, 0, torch.float32)
    assert np.abs(costs - ref_c).max() <= 2e-5 * max(1.0, np.abs(ref_c).max())
    ok = [0, 2, 3, 5]
    assert np.abs(grads[ok] - ref_g[ok]).max() <= 1e-4 and np.abs(grads - ref_g).max() <= 2e-3


def test_impossible_alignment(oracle):
    """-inf logits on the only path (the linear-domain chain ends on probability zero and hands the sample to the log-domain
    sweep): cost +inf and NaN in-lattice gradients, as the reference's arithmetic gives (tests/test_gpu_non_finite.py pins that for
    every lattice form; until round 4 the sentinel's 0.69e30 came out, with finite gradients), the sample next to it untouched."""
    rng = np.random.default_rng(10)
    acts, labels, tl, ll = _case(rng, 3, 20, 6, 5, scale=1.0, ragged=False)
    labels[0, 2] = 3
    acts[0, :, 2, 3] = -np.inf                              # sample 0 can never emit its third label
    costs, grads = run_gpu(acts, labels, tl, ll, 0, torch.float32)
    ref_c, ref_g = oracle.rnnt_logits(acts[1:2].astype(np.float64), labels[1:2], tl[1:2], ll[1:2], 0)
    full_c, full_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll, 0)
    assert np.isposinf(full_c[0]) and np.isposinf(costs[0])
    assert np.isnan(grads[0]).all() and np.isnan(full_g[0]).all() and not np.isnan(grads[1:]).any()
    assert abs(costs[1] - ref_c[0]) <= 1e-4 * max(1.0, abs(ref_c[0]))
    assert np.abs(grads[1] - ref_g[0]).max() <= 1e-4


def test_bf16_storage_uses_the_same_lattice(oracle):
    rng = np.random.default_rng(11)
    acts, labels, tl, ll = _case(rng, 3, 90, 33, 40, scale=2.0)
    q = torch.tensor(acts).to(torch.bfloat16).float().numpy()
    ref_c, ref_g, mag = oracle.rnnt_logits(q.astype(np.float64), labels, tl, ll, 0, want_mag=True)
    costs, grads = run_gpu(q, labels, tl, ll, 0, torch.bfloat16)
    assert np.abs(costs - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max())
    oracle.assert_grads(grads, ref_g, mag, torch.bfloat16)          # per element: one rounding of the stored value


def test_repeatable_bit_for_bit():
    """Ragged batch at one block per compute unit (N = 128, both directions): ten calls, identical bits.  (The operand
    wavefronts' prefetch once left loads in flight into registers the compiler had re-used: sporadic fallbacks.)"""
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    N, T, U, A = 128, 200, 41, 32
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand((N, T, U, A), generator=g, device=dev)
    labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
    tl = torch.randint(T // 2, T + 1, (N,), generator=g, device=dev, dtype=torch.int32)
    ll = torch.randint((U - 1) // 2, U, (N,), generator=g, device=dev, dtype=torch.int32)
    tl[0], ll[0] = T, U - 1
    first = None
    for i in range(10):
        costs, grads = torch.zeros(N), torch.empty_like(x)
        assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, costs, grads, 0, 0) == 0
        torch.cuda.synchronize()
        if first is None:
            first = (costs.clone(), grads.clone())
        else:
            assert torch.equal(costs, first[0]) and torch.equal(grads, first[1])
