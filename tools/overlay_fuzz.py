#!/usr/bin/env python
"""Random problems whose coefficient table is LARGER than 32 MB, i.e. whose records overlay the lattice blocks of earlier samples
(warp-transducer_amd/csrc/rnnt_host.h, make_layout): small vocabularies keep the tensors small, the lattices are big.  Every case
runs the padded entry (one stream), the padded entry with a second stream when the lattice is long enough for the two-half schedule,
the two-phase pair, and the packed entry, on ONE recycled workspace filled with garbage first; costs and gradients of every sample
against the fp64 oracle (per element, oracle.grad_bound).  Both coefficient kernel forms: tiled (maxU > 48: the in-kernel overlay
guard) and cell-per-thread (maxU <= 48: groups of samples, launch after launch).
Usage: python tools/overlay_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import torch
from oracle import oracle as O
from warprnnt_pytorch import _lib, warp_rnnt
from warprnnt_pytorch.packed import pack_joint, row_offsets

TIGHT = os.environ.get("RNNT_OVHEAD") is not None              # (dev library: the head of the workspace is ONE sample's records, the guard really waits)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
lib = _lib.lib()
O.lib().oracle_set_num_threads(min(32, os.cpu_count() or 1))
aux = torch.cuda.Stream(dev)
worst = 0.0
forms = {"tiled": 0, "cell": 0, "two_half": 0}
for it in range(cases):
    small_u = it % 3 == 2
    U = int(rng.integers(20, 49)) if small_u else int(rng.integers(49, 400))
    T = int(rng.integers(100, 1200))
    cells_needed = (33 << 20) // 16
    N = max(2, -(-cells_needed // (T * U)) + int(rng.integers(0, 6)))
    if TIGHT:                                                  # dev library with RNNT_OVHEAD=1: every table overlays, any size
        T, U = (int(rng.integers(60, 400)), int(rng.integers(20, 49))) if small_u else (int(rng.integers(60, 400)), int(rng.integers(49, 200)))
        N = int(rng.integers(3, 80))
        if it % 4 == 1:                                        # long lattices, few samples: the two-half schedule on a tight head
            T, U, N = int(rng.integers(770, 1000)), int(rng.integers(49, 90)), int(rng.integers(3, 14))
    A = int(rng.integers(2, 5))
    if N * T * U * A * 4 > 3e9:
        continue
    dtype = torch.float32 if it % 4 else torch.bfloat16
    x = torch.tensor(rng.standard_normal((N, T, U, A)).astype(np.float32) * float(rng.choice([0.5, 2.0])), device=dev).to(dtype)
    blank = int(rng.integers(0, A))
    labels = rng.integers(0, A, size=(N, U - 1)).astype(np.int32)
    labels[labels == blank] = (blank + 1) % A
    tl = rng.integers(max(1, T // 3), T + 1, size=N).astype(np.int32); tl[int(rng.integers(0, N))] = T
    ll = rng.integers(0, U, size=N).astype(np.int32); ll[int(rng.integers(0, N))] = U - 1
    assert TIGHT or 16 * N * T * U > (32 << 20)
    forms["cell" if U <= 48 else "tiled"] += 1
    ref_c, ref_g, mag = O.rnnt_logits(x.double().cpu().numpy(), labels, tl, ll, blank, want_mag=True)
    t_lab, t_tl, t_ll = (torch.tensor(v, device=dev) for v in (labels, tl, ll))
    esz = 4 if dtype == torch.float32 else 2
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, esz), dtype=torch.uint8, device=dev)
    ws.fill_(int(rng.integers(1, 255)))

    def check(got, costs, what):
        global worst
        c = costs.double().cpu().numpy()
        assert np.abs(c - ref_c).max() <= 1e-4 * max(1.0, np.abs(ref_c).max()), (it, what, N, T, U, A)
        r = O.grad_check(got.double().cpu().numpy(), ref_g, mag, dtype, rel=1e-3)
        assert r["passed"], (it, what, N, T, U, A, str(dtype), r)
        worst = max(worst, r["max_err_over_quantum"])

    for use_aux in (False, True):
        if use_aux and T + U - 1 < 768:
            continue
        if use_aux:
            warp_rnnt.set_aux_stream(aux); forms["two_half"] += 1
        try:
            costs, got = torch.zeros(N, device=dev), torch.full_like(x, 3.0)
            warp_rnnt.gpu_rnnt_async(x, t_lab, t_tl, t_ll, costs, got, blank, workspace=ws)
            torch.cuda.synchronize()
        finally:
            if use_aux:
                warp_rnnt.set_aux_stream(None)
        check(got, costs, "one-call" + (" + aux stream" if use_aux else ""))
    # two-phase pair with a per-sample scale of one
    costs, got = torch.zeros(N, device=dev), torch.full_like(x, 3.0)
    ws2 = warp_rnnt.gpu_rnnt_fwd(x, t_lab, t_tl, t_ll, costs, blank, True)
    warp_rnnt.gpu_rnnt_bwd(x, got, torch.ones(N, device=dev), ws2, blank)
    torch.cuda.synchronize()
    check(got, costs, "two-phase")
    del ws2
    # packed layout, per-sample scales (the row-scale array lives in dead lattice blocks)
    p = pack_joint(x, t_tl, t_ll).contiguous()
    offs = row_offsets(t_tl, t_ll)
    g = torch.full_like(p, 3.0)
    costs = torch.zeros(N, device=dev)
    scale = torch.ones(N, device=dev)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=blank, maxT=T, maxU=U, batch_first=True)
    st = lib.compute_rnnt_loss_packed(p.data_ptr(), g.data_ptr(), t_lab.data_ptr(), t_ll.data_ptr(), t_tl.data_ptr(), offs.data_ptr(), p.shape[0], A, N,
                                      costs.data_ptr(), scale.data_ptr(), ws.data_ptr(), opt, _lib.DT_F32 if dtype == torch.float32 else _lib.DT_BF16, 0.0)
    assert st == 0
    torch.cuda.synchronize()
    got = torch.zeros_like(x)
    o = offs.cpu().numpy()
    for b in range(N):
        got[b, :tl[b], :ll[b] + 1] = g[o[b]:o[b + 1]].view(int(tl[b]), int(ll[b]) + 1, A)
    check(got, costs, "packed")
    del x, got, g, p, ws
    torch.cuda.empty_cache()
mode = "dev library, head = ONE sample's records: the guard really waits" if TIGHT else "record table > 32 MB"
print("overlay_fuzz: %d cases (%s: %s), every sample as the oracle says in the one-call, two-half, two-phase and packed forms; "
      "worst error / bound %.2f" % (cases, mode, ", ".join("%s x%d" % kv for kv in forms.items()), worst))
