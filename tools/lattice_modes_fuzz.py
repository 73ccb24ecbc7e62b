#!/usr/bin/env python
"""Development check for the linear-domain lattice kernel: random one-wavefront problems through the DEV library
(`make -C warp-transducer_amd dev`, WARP_RNNT_PATH=.../lib/dev/libwarprnnt.so) once per RNNT_TUNE=latlin mode, results
saved; a second invocation compares the modes.
    RNNT_TUNE=latlin=0 python tools/lattice_modes_fuzz.py run out0.npz
    RNNT_TUNE=latlin=3 python tools/lattice_modes_fuzz.py run out3.npz
    python tools/lattice_modes_fuzz.py compare out0.npz out3.npz
Costs must agree to 2e-6 relative (the two kernels round differently), gradients to 2e-5 absolute where the fp32 lattice
values are O(100) (larger costs scale the bound)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
CASES = 160


def case(i):
    rng = np.random.default_rng(1000 + i)
    N = int(rng.integers(1, 40))
    T = int(rng.integers(1, 120))
    U = int(rng.integers(1, 65))
    A = int(rng.integers(2, 40))
    scale = float(rng.choice([0.3, 1.0, 3.0, 10.0, 40.0, 120.0]))
    acts = (rng.standard_normal((N, T, U, A)) * scale).astype(np.float32)
    if i % 7 == 3:
        acts[rng.random(acts.shape) < 0.05] = -np.inf
        acts[..., 0] = np.where(np.isinf(acts[..., 0]), 0.0, acts[..., 0])      # the blank stays possible
    labels = rng.integers(1, A, size=(N, max(U - 1, 1))).astype(np.int32)
    tl = rng.integers(1, T + 1, size=N).astype(np.int32)
    ll = rng.integers(0, U, size=N).astype(np.int32)
    tl[0], ll[-1] = T, U - 1
    return acts, labels, tl, ll


def run(path):
    import torch
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    out = {}
    for i in range(CASES):
        acts, labels, tl, ll = case(i)
        x = torch.tensor(acts, device=dev)
        costs = torch.zeros(x.shape[0])
        grads = torch.empty_like(x)
        rc = warp_rnnt.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev), torch.tensor(ll, device=dev),
                                costs, grads, 0, 0)
        assert rc == 0
        torch.cuda.synchronize()
        out["c%d" % i] = costs.numpy()
        out["g%d" % i] = grads.cpu().numpy()
    np.savez_compressed(path, **out)
    print("saved", path, "tune", os.environ.get("RNNT_TUNE"))


def compare(a, b):
    A, B = np.load(a), np.load(b)
    worst_c = worst_g = 0.0
    bad = 0
    for i in range(CASES):
        ca, cb, ga, gb = A["c%d" % i], B["c%d" % i], A["g%d" % i], B["g%d" % i]
        same_special = np.array_equal(np.isnan(ca), np.isnan(cb)) and np.array_equal(ca > 1e29, cb > 1e29)
        ok = np.isfinite(ca) & (ca < 1e29)
        rel = float((np.abs(ca - cb)[ok] / np.maximum(1.0, np.abs(ca[ok]))).max()) if ok.any() else 0.0
        gerr = float(np.nanmax(np.abs(ga[ok] - gb[ok]))) if ok.any() else 0.0
        bound_g = 2e-5 * max(1.0, float(np.abs(ca[ok]).max()) / 100.0) if ok.any() else 0.0
        worst_c, worst_g = max(worst_c, rel), max(worst_g, gerr / max(bound_g, 1e-30) * 2e-5)
        if not same_special or rel > 2e-6 or gerr > bound_g:
            bad += 1
            print("case %d: special-equal %s, cost rel %.3g, grad %.3g (bound %.3g), shape %s" % (i, same_special, rel, gerr, bound_g, ga.shape))
    print("%d cases, %d outside the bounds; worst cost rel %.3g, worst grad (scaled to the 2e-5 bound) %.3g" % (CASES, bad, worst_c, worst_g))
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
