#!/usr/bin/env python
"""Time the four storage types of the materialised path on one shape (development aid): ms per call of
compute_rnnt_loss_async with gradients, per-stage times from rnnt_profile_*, achieved bytes/s.
Usage: python tools/dtype_bench.py [N T U A]   (default 64 150 21 5000)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import torch
from warprnnt_pytorch import _lib, warp_rnnt

N, T, U, A = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 150, 21, 5000)
dev = torch.device("cuda:0")
lib = _lib.lib()
for dt in (torch.float32, torch.float64, torch.bfloat16, torch.float16):
    g = torch.Generator(device=dev); g.manual_seed(0)
    acts = torch.rand((N, T, U, A), generator=g, device=dev, dtype=torch.float32).to(dt)
    labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    grads = torch.empty_like(acts)
    costs = torch.empty(N, device=dev, dtype=torch.float64 if dt == torch.float64 else torch.float32)
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, acts.element_size()), dtype=torch.uint8, device=dev)

    def call():
        warp_rnnt.gpu_rnnt_async(acts, labels, tl, ll, costs, grads, 0, workspace=ws)

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.rnnt_profile_reset(); lib.rnnt_profile_enable(1)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        call()
        torch.cuda.synchronize()
        lib.rnnt_profile_collect()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    lib.rnnt_profile_enable(0)
    st = (C.c_double * 5)(); n = lib.rnnt_profile_read(st, 5)
    stages = [round(st[i] / max(n, 1), 4) for i in range(5)]
    E = N * T * U * A * acts.element_size()
    print("%-9s N=%d T=%d U=%d A=%d: %.3f ms/call, stages stats/lattice/coef/grad/span %s, stats %.2f TB/s, grad %.2f TB/s"
          % (str(dt).replace("torch.", ""), N, T, U, A, ms, stages, E / stages[0] / 1e9, 2 * E / stages[3] / 1e9))
    del acts, grads, ws
    torch.cuda.empty_cache()
