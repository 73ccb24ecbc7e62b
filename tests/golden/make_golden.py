#!/usr/bin/env python
"""Generates tests/golden/ref_cases.npz by running the REFERENCE ITSELF.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
It loads oracle/_ref/libwarprnnt_ref.so -- the reference's CPU path compiled by oracle/Makefile
from the reference sources where they lie -- and records, for a set of seeded cases, the outputs
of the reference's `compute_rnnt_loss` (fp32) and `compute_rnnt_loss_fp64` on log-probabilities:
per-sample costs and the sparse gradient wrt log-probs, plus that gradient pushed through the
log-softmax chain rule (the dense logit gradient the GPU contract returns).
Inputs are regenerated in the tests from the recorded seeds/streams; only outputs are stored.
The fixture travels to the GPU box (where /root/reference does not exist).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# name: (N, T, U, A, blank, variable_lengths, source)   source 'rng' = numpy default_rng(seed),
# 'refgen' = the reference's own generator streams (tests/random.cpp)
CASES = {
    "var_a40": (3, 17, 6, 40, 0, True, "rng"),
    "blank5_a19": (4, 11, 7, 19, 5, True, "rng"),
    "u1_t1": (3, 6, 4, 9, 2, "edge", "rng"),
    "wide_u70": (2, 33, 70, 12, 0, True, "rng"),
    "a1000": (2, 9, 4, 1000, 3, True, "rng"),
    "a5003": (1, 5, 3, 5003, 0, False, "rng"),
    "inf_test": (1, 50, 10, 15, 0, False, "refgen"),        # tests/test_cpu.cpp:181-240
    "grad_check_a20": (1, 50, 15, 20, 0, False, "refgen"),   # tests/test_cpu.cpp:347-349
    "grad_check_a5": (65, 10, 5, 5, 0, False, "refgen"),     # tests/test_cpu.cpp:350
}


def case_inputs(name):
    """Deterministic inputs of a case (shared with the tests through this module)."""
    N, T, U, A, blank, var, src = CASES[name]
    seed = sum(ord(c) for c in name)
    rng = np.random.default_rng(seed)
    if src == "refgen":
        acts = O.gen_acts(N * T * U * A).astype(np.float64).reshape(N, T, U, A)
        lab = O.gen_labels(A, U - 1)
        if name == "inf_test":
            lab[0] = 2                                        # tests/test_cpu.cpp:188
        labels = np.tile(lab, (N, 1)).astype(np.int32)
    else:
        acts = rng.standard_normal((N, T, U, A)) * 2.0
        labels = rng.integers(0, A, size=(N, U - 1)).astype(np.int32)
        labels[labels == blank] = (blank + 1) % A
    act_lens = np.full(N, T, dtype=np.int32)
    label_lens = np.full(N, U - 1, dtype=np.int32)
    if var is True:
        act_lens = rng.integers(1, T + 1, size=N).astype(np.int32)
        label_lens = rng.integers(0, U, size=N).astype(np.int32)
        act_lens[0], label_lens[-1] = T, U - 1
    elif var == "edge":                                       # U_b = 1 (empty label) and T_b = 1
        act_lens = np.array([T, 1, T], dtype=np.int32)
        label_lens = np.array([U - 1, U - 1, 0], dtype=np.int32)
    return acts, labels, act_lens, label_lens, blank


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
    out = {}
    for name in CASES:
        acts, labels, act_lens, label_lens, blank = case_inputs(name)
        lp64 = O.log_softmax(acts)
        c64, g64 = O.ref_rnnt_logprobs(lp64, labels, act_lens, label_lens, blank, True, 1)
        lp32 = O.log_softmax(acts.astype(np.float32))
        c32, g32 = O.ref_rnnt_logprobs(lp32, labels, act_lens, label_lens, blank, True, 1)
        out[name + "/costs64"] = c64
        out[name + "/lpgrad64"] = g64.astype(np.float32)      # sparse, values in [-1, 0]
        out[name + "/logitgrad64"] = O.chain_rule_to_logits(lp64, g64).astype(np.float32)
        out[name + "/costs32"] = c32
        print("%-16s cost64[0]=%.6f  |c32-c64|max=%.2e" % (name, c64[0], np.abs(c32 - c64).max()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
