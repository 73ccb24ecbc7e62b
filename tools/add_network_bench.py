#!/usr/bin/env python
"""Time the additive-joint entry against the materialised path on the BASELINE shapes.
  fused        : compute_rnnt_loss_add(f, g) -> costs, df, dg
  materialised : joint = f[:, :, None] + g[:, None]  (torch)  -> compute_rnnt_loss_async -> grads
                 -> df = grads.sum(2), dg = grads.sum(1)  (torch)   [what a user of the reference does]
  step         : RNNTLossAdd(reduction='mean') forward + backward through autograd (two-phase entry,
                 1/N and grad_output folded into the gradient kernels)
Usage: python tools/add_network_bench.py [--fused-only] [--bf16|--fp16] [c2 c3 c4 c5f32]
  --fused-only: skip the materialised comparison; --bf16 / --fp16: 16-bit storage of f, g, df, dg (fused path only)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import torch
from warprnnt_pytorch import _lib, warp_rnnt

SHAPES = {"c2": (16, 150, 41, 28), "c3": (128, 150, 21, 5000), "c4": (64, 1500, 301, 50), "c5f32": (128, 200, 41, 1024),
          "long128": (32, 1500, 301, 128), "long256": (32, 1500, 301, 256), "long1024": (16, 1500, 301, 1024)}   # long utterances, mid-size vocabularies
dev = torch.device("cuda:0")
lib = _lib.lib()
HALF = torch.bfloat16 if "--bf16" in sys.argv else (torch.float16 if "--fp16" in sys.argv else None)
FUSED_ONLY = "--fused-only" in sys.argv or HALF is not None
for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3"]:
    N, T, U, A = SHAPES[name] if name in SHAPES else tuple(int(x) for x in name.split(","))      # a named shape or N,T,U,A
    g0 = torch.Generator(device=dev); g0.manual_seed(1)
    f = torch.rand((N, T, A), generator=g0, device=dev)
    g = torch.rand((N, U, A), generator=g0, device=dev)
    if HALF is not None:
        f, g = f.to(HALF), g.to(HALF)
    labels = torch.randint(1, A, (N, U - 1), generator=g0, device=dev, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    df, dg = torch.empty_like(f), torch.empty_like(g)
    costs = torch.empty(N, device=dev)
    ws = torch.empty(_lib.workspace_bytes_add(T, U, N), dtype=torch.uint8, device=dev)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=0,
                           maxT=T, maxU=U, batch_first=True)

    code = {None: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}[HALF]

    def fused():
        if HALF is None:
            st = lib.compute_rnnt_loss_add(f.data_ptr(), g.data_ptr(), df.data_ptr(), dg.data_ptr(), labels.data_ptr(),
                                           ll.data_ptr(), tl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
            assert st == 0
            return
        st = lib.compute_rnnt_loss_add_fwd_dt(f.data_ptr(), g.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N,
                                              costs.data_ptr(), ws.data_ptr(), opt, code, 1, 0.0)
        assert st == 0
        st = lib.compute_rnnt_loss_add_bwd_dt(f.data_ptr(), g.data_ptr(), df.data_ptr(), dg.data_ptr(), None, labels.data_ptr(),
                                              ll.data_ptr(), tl.data_ptr(), A, N, ws.data_ptr(), opt, code)
        assert st == 0

    grads = None if FUSED_ONLY else torch.empty((N, T, U, A), device=dev)

    def materialised():
        joint = f.unsqueeze(2) + g.unsqueeze(1)
        warp_rnnt.gpu_rnnt_async(joint, labels, tl, ll, costs, grads, 0, workspace=ws)
        return grads.sum(2), grads.sum(1)

    if os.environ.get("RNNT_TUNE_LIVE") and os.environ.get("VARIANTS"):
        # dev library: alternate RNNT_TUNE variants INSIDE this process on the same buffers (VARIANTS="a=1;b=2,c=3;...")
        for rep in range(2):
            for v in os.environ["VARIANTS"].split(";"):
                os.environ["RNNT_TUNE"] = v
                for _ in range(3):
                    fused()
                torch.cuda.synchronize()
                lib.rnnt_profile_reset(); lib.rnnt_profile_enable(1)
                for _ in range(10):
                    fused(); torch.cuda.synchronize(); lib.rnnt_profile_collect()
                lib.rnnt_profile_enable(0)
                st = (C.c_double * 5)(); n = lib.rnnt_profile_read(st, 5)
                print("%s %-22s stages stats/lattice/coef/grad/span %s" % (name, v, [round(st[i] / max(n, 1), 4) for i in range(5)]))
        continue
    out = {}
    for label, fn in (("fused", fused),) + (() if FUSED_ONLY else (("materialised", materialised),)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.rnnt_profile_reset(); lib.rnnt_profile_enable(1)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
            lib.rnnt_profile_collect()
        ms = (time.perf_counter() - t0) * 1e3 / reps
        lib.rnnt_profile_enable(0)
        st = (C.c_double * 5)(); n = lib.rnnt_profile_read(st, 5)
        out[label] = (ms, [round(st[i] / max(n, 1), 4) for i in range(5)])
    from warprnnt_pytorch.add_network import RNNTLossAdd
    crit = RNNTLossAdd(blank=0, reduction="mean")
    fr, gr = f.clone().requires_grad_(True), g.clone().requires_grad_(True)

    def step():
        fr.grad = None; gr.grad = None
        crit(fr, gr, labels, tl, ll).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) * 1e2
    c_f = costs.clone(); fused(); torch.cuda.synchronize()
    if FUSED_ONLY:
        print("%s%s N=%d T=%d U=%d A=%d: fused %.3f ms (stages stats/lattice/coef/grad/span %s) | autograd step (mean) %.3f ms"
              % (name, "" if HALF is None else " " + str(HALF).replace("torch.", ""), N, T, U, A, out["fused"][0], out["fused"][1], step_ms))
        continue
    print("%s N=%d T=%d U=%d A=%d: fused %.3f ms (stages stats/lattice/coef/grad/span %s) | materialised %.3f ms "
          "(library stages %s) | speed-up x%.1f | autograd step (mean) %.3f ms"
          % (name, N, T, U, A, out["fused"][0], out["fused"][1], out["materialised"][0], out["materialised"][1],
             out["materialised"][0] / out["fused"][0], step_ms))
