#!/usr/bin/env python
"""Randomised check of the non-finite-logit behaviour (include/rnnt.h): random shapes, storage types, lengths and bad cells
(NaN / +inf / an all -inf row; in-lattice and padded rows; several per batch) through compute_rnnt_loss[_fp64|_bf16] on the GPU --
the samples the fp64 oracle makes NaN must be exactly the samples whose cost is NaN, the gradient NaN pattern must match, and
every other sample must equal the clean run bit for bit.  Usage: python tools/non_finite_fuzz.py [cases=200] [seed=0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from oracle import oracle as O
from warprnnt_pytorch import warp_rnnt

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)


def run(x, labels, tl, ll, ws=None):
    costs = torch.zeros(x.shape[0], dtype=x.dtype if x.dtype == torch.float64 else torch.float32)
    grads = torch.full_like(x, 3.0)
    assert warp_rnnt.gpu_rnnt(x, labels, tl, ll, costs, grads, 0, 0, workspace=ws) == 0
    torch.cuda.synchronize()
    return costs.double().numpy(), grads.double().cpu().numpy()


bad_total = 0
for case in range(cases):
    N = int(rng.integers(1, 7))
    kind = rng.integers(0, 5)
    if kind == 0:
        T, U, A = int(rng.integers(1, 40)), int(rng.integers(1, 12)), int(rng.integers(2, 60))
    elif kind == 1:
        T, U, A = int(rng.integers(2, 20)), int(rng.integers(2, 8)), int(rng.choice([1024, 1500, 3100, 5000]))
    elif kind == 2:
        T, U, A = int(rng.integers(8, 40)), int(rng.integers(64, 200)), int(rng.choice([26, 50, 52]))
    elif kind == 3:
        T, U, A = int(rng.integers(760, 820)), int(rng.integers(2, 30)), int(rng.integers(4, 40))      # two-half schedule
    else:
        T, U, A = int(rng.integers(2, 30)), int(rng.integers(2, 70)), int(rng.integers(2, 300))
    dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float64][int(rng.integers(0, 4))]
    x = torch.tensor(rng.standard_normal((N, T, U, A)).astype(np.float32), device=dev).to(dtype)
    labels = torch.tensor(rng.integers(1, A, size=(N, max(U - 1, 1))).astype(np.int32)[:, :U - 1] if U > 1 else np.zeros((N, 0), np.int32), device=dev)
    if U == 1:
        labels = torch.zeros((N, 1), dtype=torch.int32, device=dev)[:, :0].contiguous()
    tl = rng.integers(1, T + 1, size=N).astype(np.int32); tl[rng.integers(0, N)] = T
    ll = rng.integers(0, U, size=N).astype(np.int32); ll[rng.integers(0, N)] = U - 1
    t_tl, t_ll = torch.tensor(tl, device=dev), torch.tensor(ll, device=dev)
    lab_arg = labels if labels.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
    warp_rnnt.set_aux_stream(side if kind == 3 and case % 2 == 0 else None)
    c0, g0 = run(x, lab_arg, t_tl, t_ll)
    xb = x.clone()
    for _ in range(int(rng.integers(1, 4))):
        b, t, u = int(rng.integers(0, N)), int(rng.integers(0, T)), int(rng.integers(0, U))
        what = int(rng.integers(0, 3))
        if what == 2:
            xb[b, t, u, :] = float("-inf")
        else:
            xb[b, t, u, int(rng.integers(0, A))] = float("nan") if what == 0 else float("inf")
    c1, g1 = run(xb, lab_arg, t_tl, t_ll)
    warp_rnnt.set_aux_stream(None)
    rc, rg = O.rnnt_logits(xb.double().cpu().numpy(), labels.cpu().numpy().reshape(N, U - 1), tl, ll)
    bad = np.isnan(rc)
    assert np.array_equal(np.isnan(c1), bad), (case, (N, T, U, A), dtype, c1, rc)
    assert np.array_equal(np.isnan(g1), np.isnan(rg)), (case, (N, T, U, A), dtype)
    assert np.array_equal(c1[~bad], c0[~bad]) and np.array_equal(g1[~bad], g0[~bad]), (case, (N, T, U, A), dtype)
    bad_total += int(bad.sum())
print("non_finite_fuzz: %d cases, %d poisoned samples, all as the oracle says; clean samples bit-identical" % (cases, bad_total))
