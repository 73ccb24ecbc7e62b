#!/usr/bin/env python
"""The two bindings of warprnnt_pytorch side by side in ONE process (compiled extension module / ctypes), c2 shape:
host time of forward (call returns), of backward (call returns), and the whole step with one device sync.
Usage: python tools/binding_ab.py [validate=0|1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
import numpy as np
import torch
from warprnnt_pytorch import RNNTLoss, warp_rnnt

validate = not (len(sys.argv) > 1 and sys.argv[1] == "0")
EXT = warp_rnnt._EXT
dev = torch.device("cuda:0")
N, T, U, A = 16, 150, 41, 28
x = torch.rand((N, T, U, A), device=dev).requires_grad_(True)
lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
tl = torch.full((N,), T, dtype=torch.int32, device=dev)
ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
crit = RNNTLoss(reduction="mean", validate=validate)


def run(n):
    f, b, s = [], [], []
    for _ in range(n):
        x.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = crit(x, lab, tl, ll)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        f.append(t1 - t0); b.append(t2 - t1); s.append(t3 - t0)
    return [np.median(v) * 1e6 for v in (f, b, s)]


for rep in range(3):
    for name, mod in (("ext", EXT), ("ctypes", None)):
        if name == "ext" and EXT is None:
            continue
        warp_rnnt._EXT = mod
        run(50)
        f, b, s = run(300)
        print("validate=%d %-6s forward call %.1f us, backward call %.1f us, step incl. sync %.1f us" % (validate, name, f, b, s))
