#!/usr/bin/env python
"""Where the host time of a small RNNTLoss step goes: cProfile over forward + backward of the module on c2
(N=16,T=150,U=41,A=28), top functions by cumulative time.  Usage: python tools/autograd_profile.py [validate=0|1]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
import torch
from warprnnt_pytorch import RNNTLoss

validate = not (len(sys.argv) > 1 and sys.argv[1] == "0")
dev = torch.device("cuda:0")
N, T, U, A = 16, 150, 41, 28
x = torch.rand((N, T, U, A), device=dev).requires_grad_(True)
lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
tl = torch.full((N,), T, dtype=torch.int32, device=dev)
ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
try:
    crit = RNNTLoss(reduction="mean", validate=validate)
except TypeError:
    crit = RNNTLoss(reduction="mean")


def step():
    x.grad = None
    crit(x, lab, tl, ll).backward()
    torch.cuda.synchronize()


for _ in range(50):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
