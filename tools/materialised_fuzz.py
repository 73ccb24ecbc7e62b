#!/usr/bin/env python
"""Random problems through the materialised path -- fp32 / bf16 / fp16 storage, padded and packed layout, FastEmit on and off,
per-sample weights -- against this library's own FP64 path on the same (rounded) inputs (which the parity suite pins to the
oracle and to the reference): every statistics / lattice / coefficient / gradient kernel form and every dispatch edge of
rnnt_gpu.hip gets hit by chance rather than by design.  Usage: python tools/materialised_fuzz.py [cases=300] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.autograd_ref import rnnt_autograd        # fp64 autograd through an explicit lattice: the checker for label == blank
from warprnnt_pytorch import RNNTLoss
from warprnnt_pytorch.packed import RNNTLossPacked, pack_joint, unpack_joint

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst = {"cost": 0.0, "grad": 0.0}
kinds = {}
for case in range(cases):
    N = int(rng.choice([1, 2, 3, 5, 17, 40]))
    T = int(rng.choice([rng.integers(1, 60), rng.integers(1, 60), 150, 400]))
    U = int(rng.choice([rng.integers(1, 50), rng.integers(1, 50), 65, 130, 300]))
    A = int(rng.choice([rng.integers(2, 70), rng.integers(2, 70), 50, 52, 256, 1024, 1030, 5000]))
    while N * T * U * A > 30_000_000:
        if N > 1: N = max(1, N // 2)
        else: T = max(1, T // 2); U = max(1, U // 2)
    dtype = torch.float32 if rng.random() < 0.5 else (torch.bfloat16 if rng.random() < 0.5 else torch.float16)
    packed = rng.random() < 0.3
    lam = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
    scale = float(rng.choice([0.5, 1.0, 3.0, 12.0]))
    blank = int(rng.integers(0, A))
    x = torch.tensor(rng.standard_normal((N, T, U, A)) * scale, dtype=dtype, device=dev)
    if rng.random() < 0.1:
        x[torch.rand(x.shape, device=dev) < 0.03] = -float("inf")     # masked symbols
        x[..., blank] = torch.nan_to_num(x[..., blank], neginf=0.0)    # the blank stays possible
    labels = rng.integers(0, A, size=(N, max(U - 1, 0)))
    # labels EQUAL to the blank stay in (the reference's generators never draw one, tests/random.cpp:22-38): the GPU reference
    # subtracts both corrections there (gpu_rnnt_kernel.h:161-174); a third of the cases get many of them on purpose
    eqb = rng.random() < 0.33
    if eqb:
        labels[rng.random(labels.shape) < 0.4] = blank
    tl = rng.integers(1, T + 1, size=N); tl[rng.integers(0, N)] = T
    ll = rng.integers(0, U, size=N); ll[rng.integers(0, N)] = U - 1
    if rng.random() < 0.3:
        tl[:] = T; ll[:] = U - 1                       # full-length batch
    lab, ttl, tll = (torch.tensor(a.astype(np.int32), device=dev) for a in (labels, tl, ll))
    w = torch.tensor(rng.uniform(0.5, 2.0, size=N), dtype=torch.float32, device=dev)
    if os.environ.get("ONLY") and case != int(os.environ["ONLY"]):
        continue                                       # (replay one case of a seed: every random draw above still happens)
    # reference: fp64, padded layout
    xr = x.double().clone().requires_grad_(True)
    lr = RNNTLoss(blank=blank, reduction="none", fastemit_lambda=lam)(xr, lab, ttl, tll)
    (lr * w.double()).sum().backward()
    if packed:
        xp = pack_joint(x, ttl, tll).contiguous().requires_grad_(True)
        lt = RNNTLossPacked(blank=blank, reduction="none", fastemit_lambda=lam)(xp, lab, ttl, tll)
        (lt * w).sum().backward()
        got = unpack_joint(xp.grad, ttl, tll, T, U)
    else:
        xt = x.clone().requires_grad_(True)
        lt = RNNTLoss(blank=blank, reduction="none", fastemit_lambda=lam)(xt, lab, ttl, tll)
        (lt * w).sum().backward()
        got = xt.grad
    finite = torch.isfinite(lr)
    # small problems: the fp64 path itself against autograd (no gradient formula), so that label == blank is not judged by a
    # path that shares the kernels' formula
    if lam == 0.0 and N * T * U <= 3000 and bool(torch.isfinite(x).all()) and bool(finite.all()):
        ac, ag = rnnt_autograd(x.double().cpu().numpy(), labels, tl, ll, blank, w.double().cpu().numpy())
        e64c = np.abs(lr.detach().cpu().numpy() - ac).max() / max(1.0, np.abs(ac).max())
        e64g = np.abs(xr.grad.cpu().numpy() - ag).max()
        worst["fp64 vs autograd"] = max(worst.get("fp64 vs autograd", 0.0), e64g)
        kinds["(autograd-checked)"] = kinds.get("(autograd-checked)", 0) + 1
        if not (e64c <= 1e-9 and e64g <= 1e-8):
            print("MISMATCH case %d (fp64 path vs fp64 autograd): N=%d T=%d U=%d A=%d blank=%d labels==blank: %d: cost %.2e grad %.2e"
                  % (case, N, T, U, A, blank, int((labels == blank).sum()), e64c, e64g))
            sys.exit(1)
    ec = float(((lt.detach().double() - lr.detach()).abs() / lr.detach().abs().clamp_min(1.0))[finite].max()) if finite.any() else 0.0
    same_inf = bool((torch.isfinite(lt.detach()) == finite).all())
    quant = 0.0 if dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)     # half an ulp of the stored gradient
    ref = xr.grad
    mask = finite.view(-1, 1, 1, 1).expand_as(ref)
    err = ((got.double() - ref).abs() - quant * ref.abs())[mask]
    eg = float(err.max()) if err.numel() else 0.0
    tol_g = 2.0 * 1e-3 * (1.0 + 0.0)                   # north_star: 1e-3 absolute (x the largest per-sample weight)
    if T + U > 300: tol_g *= 3.0                        # long lattices: the fp32 lattice's own rounding (5e-3 at c4's size)
    tol_g *= max(1.0, scale / 3.0)                     # logits of magnitude 30+: their fp32 rounding
    key = "%s%s%s" % (str(dtype).replace("torch.", ""), " packed" if packed else "", " fastemit" if lam else "")
    kinds[key] = kinds.get(key, 0) + 1
    worst["cost"] = max(worst["cost"], ec); worst["grad"] = max(worst["grad"], eg / tol_g)
    if not (ec <= 1e-4 and eg <= tol_g and same_inf):
        print("MISMATCH case %d: N=%d T=%d U=%d A=%d %s packed=%s lam=%g scale=%g blank=%d tl=%s ll=%s: cost %.2e grad %.2e (tol %.1e) inf-pattern %s"
              % (case, N, T, U, A, dtype, packed, lam, scale, blank, tl, ll, ec, eg, tol_g, same_inf))
        print("costs  got:", lt.detach().cpu().numpy())
        print("costs  ref:", lr.detach().cpu().numpy())
        sys.exit(1)
print("%d cases agree with the fp64 path: worst relative cost error %.2e, worst gradient error %.2f of the bound; fp64 path vs autograd "
      "worst gradient error %.1e; %s"
      % (cases, worst["cost"], worst["grad"], worst.get("fp64 vs autograd", 0.0), ", ".join("%s x%d" % kv for kv in sorted(kinds.items()))))
