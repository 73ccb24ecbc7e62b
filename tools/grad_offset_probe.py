#!/usr/bin/env python3
"""Does the gradient kernel's duration depend on WHERE the gradient tensor lies relative to the activations?
One process, one allocation holding both: activations at 0, gradients at (bytes of the activations + gap); the gradient stage
(library HIP-event timers) per gap, plus separately allocated tensors as the caller normally has them.
    python tools/grad_offset_probe.py [c3|c5] [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import torch
from warprnnt_pytorch import _lib

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, T, U, A, tdt = {"c3": (128, 150, 21, 5000, torch.float32), "c5": (128, 200, 41, 1024, torch.bfloat16)}[wl]
lib = _lib.lib()
fn = lib.compute_rnnt_loss if tdt == torch.float32 else lib.compute_rnnt_loss_bf16
dev = torch.device("cuda:0")
esz = 4 if tdt == torch.float32 else 2
nbytes = N * T * U * A * esz
SLACK = 1 << 30
pool = torch.empty(2 * nbytes + SLACK, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev); g.manual_seed(3)
acts = pool[:nbytes].view(tdt).view(N, T, U, A)
acts.copy_(torch.randn((N, T, U, A), generator=g, device=dev, dtype=torch.float32).to(tdt) if esz == 2 else torch.randn((N, T, U, A), generator=g, device=dev))
labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
tl = torch.full((N,), T, dtype=torch.int32, device=dev)
ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.zeros(N, dtype=torch.float32)
ws = torch.empty(_lib.workspace_bytes(T, U, N, True, esz), dtype=torch.uint8, device=dev)
opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=0,
                       maxT=T, maxU=U, batch_first=True)


def run(grads, what):
    argv = (acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)

    def step():
        st = fn(*argv)
        assert st == 0, _lib.status_string(st)
        lib.rnnt_profile_collect()
    for _ in range(2):
        step()
    lib.rnnt_profile_reset()
    lib.rnnt_profile_enable(1)
    for _ in range(steps):
        step()
    lib.rnnt_profile_enable(0)
    stage = (C.c_double * 5)()
    calls = lib.rnnt_profile_read(stage, 5)
    print("%-44s statistics %.4f  gradient %.4f ms   (acts %#x, grads %#x)" % (what, stage[0] / calls, stage[3] / calls, acts.data_ptr(), grads.data_ptr()), flush=True)


for gap in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 33 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20, (512 << 20) + (2 << 20), 0):
    grads = pool[nbytes + gap: 2 * nbytes + gap].view(tdt).view(N, T, U, A)
    run(grads, "one allocation, gap %d KB" % (gap >> 10))
for i in range(3):
    sep = torch.empty((N, T, U, A), dtype=tdt, device=dev)
    run(sep, "separate allocation #%d" % i)
    keep = sep if i == 0 else None
run(acts, "in place")
