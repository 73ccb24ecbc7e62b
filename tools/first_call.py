"""Cold start: what the first GPU call of a process costs (python tools/first_call.py [x]; with an argument a 1x1x1
problem goes first, so the code-object load is separated from the c2-sized call behind it).
python tools/first_call.py add | add16: the same for the additive joint (fp32 / bf16 storage): the first compute_rnnt_loss_add_fwd_dt +
_bwd_dt pair of the process against the ones behind it (each storage type has a code object of its own)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "warp-transducer_amd"))
import torch
from warprnnt_pytorch import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
N, T, U, A = 16, 150, 41, 28
x = torch.rand((N, T, U, A), device=dev)
lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
tl = torch.full((N,), T, dtype=torch.int32, device=dev); ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
g = torch.empty_like(x); costs = torch.zeros(N)
ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=0, maxT=T, maxU=U, batch_first=True)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1].startswith("add"):
    half = sys.argv[1] == "add16"
    dt, code = (torch.bfloat16, _lib.DT_BF16) if half else (torch.float32, _lib.DT_F32)
    f = torch.rand((N, T, 512), device=dev).to(dt); gg = torch.rand((N, U, 512), device=dev).to(dt)
    df, dg = torch.empty_like(f), torch.empty_like(gg)
    cd = torch.zeros(N, device=dev)
    wsa = torch.empty(_lib.workspace_bytes_add(T, U, N), dtype=torch.uint8, device=dev)
    lab512 = torch.randint(1, 512, (N, U - 1), device=dev, dtype=torch.int32)
    torch.cuda.synchronize()
    for i in range(4):
        t0 = time.perf_counter()
        st = lib.compute_rnnt_loss_add_fwd_dt(f.data_ptr(), gg.data_ptr(), lab512.data_ptr(), ll.data_ptr(), tl.data_ptr(), 512, N, cd.data_ptr(),
                                              wsa.data_ptr(), opt, code, 1, 0.0)
        st |= lib.compute_rnnt_loss_add_bwd_dt(f.data_ptr(), gg.data_ptr(), df.data_ptr(), dg.data_ptr(), None, lab512.data_ptr(), ll.data_ptr(),
                                               tl.data_ptr(), 512, N, wsa.data_ptr(), opt, code)
        torch.cuda.synchronize()
        print("additive joint (%s) call %d: %.3f ms (status %d)" % ("bf16" if half else "fp32", i, (time.perf_counter() - t0) * 1e3, st))
    sys.exit(0)
if len(sys.argv) > 1:      # a 1x1x1 problem first: what it costs is the code-object load (+ the staging buffer), not the problem
    x1 = torch.rand((1, 1, 1, A), device=dev); g1 = torch.empty_like(x1); c1 = torch.zeros(1)
    one = torch.ones(1, dtype=torch.int32, device=dev); zero = torch.zeros(1, dtype=torch.int32, device=dev)
    o1 = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=0, maxT=1, maxU=1, batch_first=True)
    w1 = torch.empty(_lib.workspace_bytes(1, 1, 1, True, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = lib.compute_rnnt_loss(x1.data_ptr(), g1.data_ptr(), zero.data_ptr(), zero.data_ptr(), one.data_ptr(), A, 1, c1.data_ptr(), w1.data_ptr(), o1)
    print("call on a 1x1x1 problem: %.3f ms (status %d)" % ((time.perf_counter() - t0) * 1e3, st))
    t0 = time.perf_counter(); p = torch.empty(1024, pin_memory=True); print("call torch pinned alloc: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
for i in range(4):
    t0 = time.perf_counter()
    st = lib.compute_rnnt_loss(x.data_ptr(), g.data_ptr(), lab.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
    t1 = time.perf_counter()
    print("call %d: %.3f ms (status %d)" % (i, (t1 - t0) * 1e3, st))
