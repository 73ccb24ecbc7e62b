#!/usr/bin/env python
"""RNNTLoss through autograd (the reference's pytorch_binding/test/test_time.py role): forward + backward of the module on the
BASELINE shapes, wall clock per step with one device sync per step, beside the C-ABI call of bench.py.
Usage: python tools/autograd_bench.py [c2 c3 c5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
import numpy as np
import torch
from warprnnt_pytorch import RNNTLoss, warp_rnnt
print("binding: %s (WARPRNNT_BINDING=ctypes selects the ctypes loader)" % warp_rnnt.binding())

SHAPES = {"c2": (16, 150, 41, 28, torch.float32), "c3": (128, 150, 21, 5000, torch.float32), "c4": (64, 1500, 301, 50, torch.float32),
          "c5": (128, 200, 41, 1024, torch.bfloat16)}
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["c2", "c3", "c5"]:
    N, T, U, A, dt = SHAPES[name]
    x = torch.rand((N, T, U, A), device=dev).to(dt).requires_grad_(True)
    lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    crit = RNNTLoss(reduction="mean")
    def step():
        x.grad = None
        loss = crit(x, lab, tl, ll)
        loss.backward()
        torch.cuda.synchronize()
    for _ in range(10):
        step()
    ts = []
    for _ in range(100 if name == "c2" else 30):
        t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s N=%d T=%d U=%d A=%d %s: RNNTLoss forward + backward median %.4f ms (p10 %.4f, p90 %.4f)"
          % (name, N, T, U, A, str(dt).replace("torch.", ""), np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
    if name == "c2":       # the same step without the device-to-host read of the lengths (RNNTLoss(validate=False))
        crit = RNNTLoss(reduction="mean", validate=False)
        for _ in range(10):
            step()
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%s validate=False: RNNTLoss forward + backward median %.4f ms (p10 %.4f, p90 %.4f)"
              % (name, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
        # ... and the whole step (forward + backward) captured in a HIP graph: step = replay + device sync
        def enqueue():
            x.grad = None
            crit(x, lab, tl, ll).backward()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                enqueue()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            enqueue()
        for _ in range(10):
            graph.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%s validate=False, step captured in a HIP graph: replay median %.4f ms (p10 %.4f, p90 %.4f)"
              % (name, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
