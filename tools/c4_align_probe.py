#!/usr/bin/env python
"""c4 statistics kernel vs ADDRESS layout: the same binary measured 0.90 or 1.09 ms depending on the process (EXPERIMENTS 11).
One process, the same inputs, the workspace and the activations placed at different offsets inside over-allocated buffers:
stage times of compute_rnnt_loss_async per placement.  Usage: python tools/c4_align_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))
import numpy as np, torch
from warprnnt_pytorch import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
N, T, U, A = 64, 1500, 301, 50
DT, ES, TDT = 0, 4, torch.float32                      # PROBE_SHAPE=N,T,U,A[,bf16]: another workload (default: c4, fp32)
if os.environ.get("PROBE_SHAPE"):
    parts = os.environ["PROBE_SHAPE"].split(",")
    N, T, U, A = (int(x) for x in parts[:4])
    if len(parts) > 4 and parts[4] == "bf16":
        DT, ES, TDT = 2, 2, torch.bfloat16
E = N * T * U * A
PAD = 8 << 20
raw = torch.empty(E * ES + 2 * PAD, dtype=torch.uint8, device=dev)
graw = torch.empty(E * ES + 2 * PAD, dtype=torch.uint8, device=dev)
lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
tl = torch.full((N,), T, dtype=torch.int32, device=dev)
ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.empty(N, device=dev)
wsb = _lib.workspace_bytes(T, U, N, True, ES)
wraw = torch.empty(wsb + 2 * PAD, dtype=torch.uint8, device=dev)
print("base addresses: acts %#x grads %#x ws %#x" % (raw.data_ptr(), graw.data_ptr(), wraw.data_ptr()))
src = torch.rand(E // 8, device=dev).to(TDT)
stream = torch.cuda.current_stream().cuda_stream

def run(aoff, woff, reps=6):
    a = raw[aoff:aoff + E * ES].view(TDT)
    for i in range(8):
        a[i * (E // 8):(i + 1) * (E // 8)].copy_(src)
    g = graw[aoff:aoff + E * ES].view(TDT)
    ws = wraw[woff:woff + wsb]
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=stream, blank_label=0, maxT=T, maxU=U, batch_first=True)
    lib.rnnt_profile_reset(); lib.rnnt_profile_enable(1)
    for i in range(reps + 2):
        if i == 2:
            torch.cuda.synchronize(); lib.rnnt_profile_collect(); lib.rnnt_profile_reset()
        st = lib.compute_rnnt_loss_async(a.data_ptr(), g.data_ptr(), lab.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N,
                                         costs.data_ptr(), None, ws.data_ptr(), opt, DT)
        assert st == 0
        torch.cuda.synchronize(); lib.rnnt_profile_collect()
    ms = (ctypes.c_double * 5)()
    calls = lib.rnnt_profile_read(ms, 5)
    lib.rnnt_profile_enable(0)
    return [m / max(calls, 1) for m in ms]

if os.environ.get("RNNT_TUNE_LIVE"):       # dev library: alternate kernel variants inside this one process, same buffers
    for rep in range(3):
        for v in sys.argv[1:] or ["tile2d=1", "tile2d=2"]:
            os.environ["RNNT_TUNE"] = v
            m = run(0, 0)
            print("%-16s statistics %.4f  lattice %.4f  coefficients %.4f  gradient %.4f  (ms)" % (v, m[0], m[1], m[2], m[3]))
    sys.exit(0)
if os.environ.get("PROBE_FAR"):            # large relative shifts of the workspace (64 MB ... 1 GB) against the activations
    BIG = 1 << 30
    wbig = torch.empty(wsb + BIG + PAD, dtype=torch.uint8, device=dev)
    wraw = wbig
    for woff in [0, 1 << 26, 1 << 27, 3 << 26, 1 << 28, 5 << 26, 1 << 29, (1 << 29) + (1 << 26), 7 << 27, 1 << 30, 0]:
        m = run(0, woff)
        print("ws +%-11d statistics %.4f  lattice %.4f  coefficients %.4f  gradient %.4f  (ms)" % (woff, m[0], m[1], m[2], m[3]))
    sys.exit(0)
for aoff, woff in [(0, 0), (0, 256), (0, 4096), (0, 65536), (0, 1 << 20), (0, (2 << 20) + 256), (16, 0), (4096, 0), (65536, 0), (1 << 20, 0),
                   ((1 << 20) + 4096, 4096), (0, 0)]:
    m = run(aoff, woff)
    print("acts +%-8d ws +%-8d  statistics %.4f  lattice %.4f  coefficients %.4f  gradient %.4f  (ms)" % (aoff, woff, m[0], m[1], m[2], m[3]))
