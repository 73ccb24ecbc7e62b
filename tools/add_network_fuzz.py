#!/usr/bin/env python
"""Random shapes through the additive-joint loss against this library's own materialised path in FP64 on the same GPU (which the
parity suite pins to the oracle): every dispatch edge of rnnt_joint.hip -- small / sampled Z kernels, cell / tiled coefficients, one-hot /
epilogue corrections, split contractions, columns per lane, 16-bit storage, far cells -- gets hit by chance rather than by design.
Usage: python tools/add_network_fuzz.py [cases=300] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
import numpy as np, torch
from warprnnt_pytorch import RNNTLoss
from warprnnt_pytorch.add_network import RNNTLossAdd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst = {"cost": 0.0, "df": 0.0, "dg": 0.0}
floor = 0
for case in range(cases):
    N = int(rng.integers(1, 4))
    T = int(rng.choice([rng.integers(1, 90), rng.integers(1, 90), 520, 600]))
    U = int(rng.choice([rng.integers(1, 90), rng.integers(1, 90), 130, 300]))
    A = int(rng.choice([rng.integers(2, 70), rng.integers(2, 300), 128, 256, 512, 1030]))
    while N * T * U * A > 40_000_000:
        T = max(1, T // 2); U = max(1, U // 2)
    dtype = torch.float32 if rng.random() < 0.6 else (torch.bfloat16 if rng.random() < 0.5 else torch.float16)
    blank = int(rng.integers(0, A))
    f = torch.tensor(rng.standard_normal((N, T, A)) * 1.5, dtype=dtype, device=dev)
    g = torch.tensor(rng.standard_normal((N, U, A)) * 1.5, dtype=dtype, device=dev)
    far = rng.random() < 0.15
    if far:                                            # rows peaking far apart: direct branches, far cells
        f[int(rng.integers(0, N)), ::3, int(rng.integers(0, A))] += 70.0
        g[int(rng.integers(0, N)), ::2, int(rng.integers(0, A))] += 90.0
    if rng.random() < 0.12 and A > 2:                  # masked symbols: -inf in f (all u of a time step) or g (all t of a label row)
        f[torch.rand(f.shape, device=dev) < 0.04] = -float("inf")
        g[torch.rand(g.shape, device=dev) < 0.04] = -float("inf")
        f[..., blank] = torch.nan_to_num(f[..., blank], neginf=0.0)     # the blank stays possible
        g[..., blank] = torch.nan_to_num(g[..., blank], neginf=0.0)
    labels = rng.integers(0, A, size=(N, max(U - 1, 0)))
    # labels EQUAL to the blank stay in (round 5: both corrections then fall on the blank column, gpu_rnnt_kernel.h:161-174; the fp64
    # materialised path this sweep compares with is pinned against autograd for that case: tests/test_gpu_label_equals_blank.py);
    # a third of the cases get many of them on purpose
    if rng.random() < 0.33:
        labels[rng.random(labels.shape) < 0.4] = blank
    tl = rng.integers(1, T + 1, size=N); tl[rng.integers(0, N)] = T
    ll = rng.integers(0, U, size=N); ll[rng.integers(0, N)] = U - 1
    lab, ttl, tll = (torch.tensor(a.astype(np.int32), device=dev) for a in (labels, tl, ll))
    w = torch.tensor(rng.uniform(0.5, 2.0, size=N), dtype=torch.float32, device=dev)
    if os.environ.get("ONLY") and case != int(os.environ["ONLY"]):
        continue                                       # (replay one case of a seed: every random draw above still happens)
    fa, ga = f.clone().requires_grad_(True), g.clone().requires_grad_(True)
    la = RNNTLossAdd(blank=blank, reduction="none")(fa, ga, lab, ttl, tll)
    (la * w).sum().backward()
    # reference: this library's materialised path in FP64 on the (rounded) inputs
    fm, gm = f.double().clone().requires_grad_(True), g.double().clone().requires_grad_(True)
    joint = (fm.unsqueeze(2) + gm.unsqueeze(1)).contiguous()
    lm = RNNTLoss(blank=blank, reduction="none")(joint, lab, ttl, tll)
    (lm * w.double()).sum().backward()
    fin = torch.isfinite(lm.detach())
    same = bool((torch.isfinite(la.detach()) == fin).all())
    ec = float(((la.detach().double() - lm.detach()).abs() / lm.detach().abs().clamp_min(1.0))[fin].max()) if fin.any() else 0.0
    if not same:
        print("MISMATCH case %d: finite / infinite costs differ: %s vs %s" % (case, la.detach().cpu().numpy(), lm.detach().cpu().numpy())); sys.exit(1)
    if not fin.all():                                  # (samples without an alignment: +inf on both sides, NaN gradients; compare the others)
        keep = fin.view(-1, 1, 1)
        for t_ in (fa.grad, ga.grad, fm.grad, gm.grad):
            t_.masked_fill_(~keep.expand_as(t_), 0.0)
    quant = 0.0 if dtype == torch.float32 else (8e-3 if dtype == torch.bfloat16 else 1e-3)    # storage quantum of 16-bit gradients (relative)
    def excess(a, b, cells):       # tests/test_gpu_add_network.py: |err| <= 2e-4 max(1, cells / 32) + 5e-5 |ref|  (x 3: per-sample weights up to 2, and the bound is what the fp32 materialised path itself just meets)
        a = a.double()
        bound = 3.0 * (2e-4 * max(1.0, cells / 32) + 5e-5 * b.abs()) + quant * b.abs().clamp_min(1.0)
        if far and quant: bound = bound + 4 * quant * b.abs().clamp_min(1.0)   # far cells reach a 16-bit gradient in up to ceil(U / 64) / ceil(T / 64) rounded additions
        if far: bound = bound + 2e-3 * b.abs().clamp_min(1.0)   # logits of magnitude 100+: their fp32 rounding alone (the materialised fp32 path shows the same)
        return float(((a - b).abs() / bound).max())
    edf, edg = excess(fa.grad, fm.grad, U), excess(ga.grad, gm.grad, T)
    bad = not (ec <= 2e-4 and edf <= 1.0 and edg <= 1.0)
    worst["cost"] = max(worst["cost"], ec)
    if dtype == torch.float32:
        worst["df"] = max(worst["df"], edf); worst["dg"] = max(worst["dg"], edg)
    if bad:
        # beyond the bound: is it the fp32 arithmetic itself?  The fp32 MATERIALISED path against the same fp64 reference
        f3, g3 = f.float().clone().requires_grad_(True), g.float().clone().requires_grad_(True)
        l3 = RNNTLoss(blank=blank, reduction="none")((f3.unsqueeze(2) + g3.unsqueeze(1)).contiguous(), lab, ttl, tll)
        (l3 * w).sum().backward()
        mdf, mdg = excess(f3.grad, fm.grad, U), excess(g3.grad, gm.grad, T)
        if ec <= 2e-4 and edf <= 1.5 * mdf + 0.5 and edg <= 1.5 * mdg + 0.5:
            floor += 1                                 # long lattices / logits of magnitude 100+: both fp32 paths sit at the same distance
            continue
        print("far=%s; fp32 materialised path: df %.2f dg %.2f of the bound" % (far, mdf, mdg))
        if dtype != torch.float32:                     # the same (rounded) inputs through the additive joint with fp32 storage
            f4, g4 = f.float().clone().requires_grad_(True), g.float().clone().requires_grad_(True)
            l4 = RNNTLossAdd(blank=blank, reduction="none")(f4, g4, lab, ttl, tll)
            (l4 * w).sum().backward()
            print("additive joint, fp32 storage: df %.2f dg %.2f of the bound" % (excess(f4.grad, fm.grad, U), excess(g4.grad, gm.grad, T)))
            d = (ga.grad.double() - gm.grad).abs(); i = int(d.argmax())
            print("worst dg element: 16-bit %.6f  fp32-storage %.6f  fp64 reference %.6f" % (float(ga.grad.flatten()[i]), float(g4.grad.flatten()[i]), float(gm.grad.flatten()[i])))
        print("MISMATCH case %d: N=%d T=%d U=%d A=%d %s blank=%d tl=%s ll=%s  cost %.2e df %.2f dg %.2f of the bound" % (case, N, T, U, A, dtype, blank, tl, ll, ec, edf, edg))
        sys.exit(1)
print("%d cases agree with the fp64 materialised path (%d of them beyond the test suite's bound together with the fp32 materialised path); "
      "worst (fp32): relative cost error %.2e, df / dg errors at %.2f / %.2f of the bound" % (cases, floor, worst["cost"], worst["df"], worst["dg"]))
