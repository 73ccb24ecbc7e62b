#!/usr/bin/env python
"""End-to-end PyTorch training-step time of the loss: RNNTLoss(reduction='mean') forward + backward
on a (N,T,U,A) logits tensor that requires grad.  Two routes of the wrapper:
  two-phase : compute_rnnt_loss_fwd in forward (keeps only the workspace), compute_rnnt_loss_bwd in
              backward with 1/N and grad_output folded into the gradient kernel        (default)
  reference : gradients computed in forward, kept in ctx, divided by N, multiplied by grad_output in
              backward -- the reference binding's flow (WARPRNNT_SYNC_API=1)
  packed    : (--varlen only) the same ragged batch in the packed layout, RNNTLossPacked (no padded rows at all)
Usage: python tools/train_step_bench.py [--varlen] [c3 c5 ...]   (--varlen: T_b ~ U[T/2,T], L_b ~ U[L/2,L])"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import torch
import warprnnt_pytorch
from warprnnt_pytorch import RNNTLoss

SHAPES = {"c2": (16, 150, 41, 28, torch.float32), "c3": (128, 150, 21, 5000, torch.float32),
          "c4": (64, 1500, 301, 50, torch.float32), "c5": (128, 200, 41, 1024, torch.bfloat16)}
dev = torch.device("cuda:0")
VARLEN = "--varlen" in sys.argv
for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3"]:
    N, T, U, A, dt = SHAPES[name]
    g = torch.Generator(device=dev); g.manual_seed(3)
    acts = torch.rand((N, T, U, A), generator=g, device=dev).to(dt).requires_grad_(True)
    labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    if VARLEN:
        tl = torch.randint(T // 2, T + 1, (N,), generator=g, device=dev, dtype=torch.int32); tl[0] = T
        ll = torch.randint((U - 1) // 2, U, (N,), generator=g, device=dev, dtype=torch.int32); ll[0] = U - 1
    res = {}
    for route, flag in (("two-phase", True), ("reference-flow", False)):
        warprnnt_pytorch._ASYNC_GPU = flag
        fn = RNNTLoss(reduction="mean")
        for _ in range(3):
            acts.grad = None
            fn(acts, labels, tl, ll).backward()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            acts.grad = None
            loss = fn(acts, labels, tl, ll)
            loss.backward()
        torch.cuda.synchronize()
        res[route] = ((time.perf_counter() - t0) * 1e3 / reps, (torch.cuda.max_memory_allocated() - base) / 2**30,
                      float(loss.detach()))
    if VARLEN:
        from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint
        warprnnt_pytorch._ASYNC_GPU = True
        pk = pack_joint(acts.detach(), tl, ll).contiguous().requires_grad_(True)
        fn = RNNTLossPacked(reduction="mean")
        for _ in range(3):
            pk.grad = None
            fn(pk, labels, tl, ll, max_T=T, max_U=U).backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pk.grad = None
            loss = fn(pk, labels, tl, ll, max_T=T, max_U=U)
            loss.backward()
        torch.cuda.synchronize()
        print("%s variable lengths, packed layout: %.3f ms/step ; loss %.6f" % (name, (time.perf_counter() - t0) * 100, float(loss.detach())))
        del pk
    a, b = res["two-phase"], res["reference-flow"]
    print("%s N=%d T=%d U=%d A=%d %s: two-phase %.3f ms/step (peak extra %.2f GiB) | reference flow %.3f ms/step "
          "(peak extra %.2f GiB) | x%.2f ; loss %.6f vs %.6f" % (name, N, T, U, A, str(dt).split(".")[-1], a[0], a[1],
                                                               b[0], b[1], b[0] / a[0], a[2], b[2]))
