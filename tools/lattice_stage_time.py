#!/usr/bin/env python3
"""Stage times of one loss + gradient call for an arbitrary shape and storage type (the library's own HIP-event timers):
    python tools/lattice_stage_time.py fp64 16,1500,301,50 [steps]
WARP_RNNT_PATH selects another library (with WARPRNNT_BINDING=ctypes), e.g. a build of an earlier commit for an A/B."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
    sys.path.insert(0, p)
import torch
from warprnnt_pytorch import _lib

dt = sys.argv[1]
N, T, U, A = (int(x) for x in sys.argv[2].split(","))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tdt = {"fp32": torch.float32, "fp64": torch.float64, "bf16": torch.bfloat16, "fp16": torch.float16}[dt]
lib = _lib.lib()
fn = {"fp32": lib.compute_rnnt_loss, "fp64": lib.compute_rnnt_loss_fp64, "bf16": lib.compute_rnnt_loss_bf16, "fp16": lib.compute_rnnt_loss_fp16}[dt]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
acts = torch.randn((N, T, U, A), generator=g, device=dev, dtype=torch.float32).to(tdt)
grads = torch.empty_like(acts)
labels = torch.randint(1, A, (N, U - 1), generator=g, device=dev, dtype=torch.int32)
tl = torch.full((N,), T, dtype=torch.int32, device=dev)
ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.zeros(N, dtype=torch.float64 if dt == "fp64" else torch.float32)
ws = torch.empty(_lib.workspace_bytes(T, U, N, True, acts.element_size()), dtype=torch.uint8, device=dev)
opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=0,
                       maxT=T, maxU=U, batch_first=True)
argv = (acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)


def step():
    st = fn(*argv)
    assert st == 0, _lib.status_string(st)
    lib.rnnt_profile_collect()


for _ in range(3):
    step()
lib.rnnt_profile_reset()
lib.rnnt_profile_enable(1)
for _ in range(steps):
    step()
lib.rnnt_profile_enable(0)
stage = (C.c_double * 5)()
calls = lib.rnnt_profile_read(stage, 5)
print("%s N=%d T=%d U=%d A=%d: statistics %.4f lattice %.4f coefficients %.4f gradient %.4f ms; cost[0] %.6f"
      % (dt, N, T, U, A, stage[0] / calls, stage[1] / calls, stage[2] / calls, stage[3] / calls, float(costs[0])))
