"""Pins the oracle (oracle/rnnt_oracle.c) to the reference: its literal golden vectors, the
committed outputs of the reference itself (tests/golden/ref_cases.npz) and, when oracle/_ref is
present, the reference CPU library called live."""
import numpy as np
import pytest

from tests.golden import literals as G
from tests.golden.make_golden import CASES, case_inputs

FIX = np.load(__file__.replace("test_oracle.py", "golden/ref_cases.npz"))


def test_small_test_cost_and_grads(oracle):
    # tests/test_cpu.cpp:26,65-70 (cost +-1e-4); pytorch_binding/test/test.py:61-78 (dense grads)
    for dt, tol in ((np.float32, 1e-6), (np.float64, 1e-7)):
        c, g = oracle.rnnt_logits(G.SMALL_ACTS.astype(dt), G.SMALL_LABELS, [2], [2])
        assert abs(c[0] - G.SMALL_COST) < 1e-4
        assert np.abs(g - G.SMALL_GRADS).max() < max(tol, 2e-8) + 1e-7


def test_options_test_cpu_contract(oracle):
    # tests/test_cpu.cpp:73-179: log-probs in, sparse log-prob grads out, eps 1e-4
    lp = oracle.log_softmax(G.OPTIONS_ACTS_6DP.astype(np.float32))
    c, g = oracle.rnnt_logprobs(lp, G.OPTIONS_LABELS, [4, 4], [2, 2])
    assert np.abs(c - G.OPTIONS_COSTS).max() < 1e-4
    assert np.abs(g - G.OPTIONS_LOGPROB_GRADS).max() < 1e-4


def test_options_test_gpu_contract(oracle):
    # tests/test_gpu.cu:117-133: raw acts in, dense logit grads out, eps 1e-4
    c, g = oracle.rnnt_logits(G.OPTIONS_ACTS_6DP.astype(np.float32), G.OPTIONS_LABELS, [4, 4], [2, 2])
    assert np.abs(c - G.OPTIONS_COSTS).max() < 1e-4
    assert np.abs(g - G.OPTIONS_LOGIT_GRADS_6DP).max() < 1e-4


def test_big_test_full_precision(oracle):
    # pytorch_binding/test/test.py:83-161 (sum of costs; grads rtol 1e-3)
    c, g = oracle.rnnt_logits(G.BIG_ACTS, G.OPTIONS_LABELS, [4, 4], [2, 2])
    assert np.allclose(c.sum(), G.OPTIONS_COSTS.sum())
    assert np.allclose(g, G.BIG_GRADS, rtol=1e-3, atol=1e-8)


@pytest.mark.parametrize("name", sorted(CASES))
def test_against_reference_fixture(oracle, name):
    acts, labels, act_lens, label_lens, blank = case_inputs(name)
    lp = oracle.log_softmax(acts)
    c, g = oracle.rnnt_logprobs(lp, labels, act_lens, label_lens, blank)
    assert np.abs(c - FIX[name + "/costs64"]).max() < 1e-9 * max(1, np.abs(c).max())
    assert np.abs(g - FIX[name + "/lpgrad64"]).max() < 1e-6          # fixture stored as fp32
    c2, g2 = oracle.rnnt_logits(acts, labels, act_lens, label_lens, blank)
    assert np.abs(c2 - c).max() < 1e-9 * max(1, np.abs(c).max())
    assert np.abs(g2 - FIX[name + "/logitgrad64"]).max() < 1e-6
    c32, _ = oracle.rnnt_logprobs(oracle.log_softmax(acts.astype(np.float32)), labels, act_lens, label_lens, blank)
    assert np.abs(c32 - FIX[name + "/costs32"]).max() < 2e-4
    # padded region exactly zero (reference behaviour probed in SURVEY.md section 4)
    for b in range(acts.shape[0]):
        assert not g2[b, act_lens[b]:].any() and not g2[b, :, label_lens[b] + 1:].any()


@pytest.mark.parametrize("name", ["var_a40", "u1_t1", "inf_test"])
def test_against_reference_live(oracle, name):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    acts, labels, act_lens, label_lens, blank = case_inputs(name)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        lp = oracle.log_softmax(acts.astype(dt))
        c, g = oracle.rnnt_logprobs(lp, labels, act_lens, label_lens, blank)
        cr, gr = oracle.ref_rnnt_logprobs(lp, labels, act_lens, label_lens, blank)
        assert np.abs(c - cr).max() <= tol * max(1, np.abs(cr).max())
        assert np.abs(g - gr).max() <= tol


def test_generators_match_reference_streams(oracle):
    # first draws of mt19937(0)/uniform(0,1) and the label stream of tests/random.cpp
    a = oracle.gen_acts(4)
    assert np.allclose(a, [0.5928446, 0.84426576, 0.8579456, 0.8472517], atol=1e-7)
    lab = oracle.gen_labels(28, 40)
    assert lab.min() >= 1 and lab.max() <= 27
    assert lab[20] == lab[21] and lab[19] == lab[20]          # forced repeats (random.cpp:33-36)
