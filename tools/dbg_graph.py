import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "warp-transducer_amd"))
import numpy as np, torch
from tests.test_gpu_add_network import problem
from warprnnt_pytorch import _lib
from oracle import oracle
f, g, labels, tl, ll, blank = problem((3, 25, 8, 96), 21)
dev = torch.device("cuda:0")
N, T, A = f.shape; U = g.shape[1]
tf, tg = torch.tensor(f, device=dev), torch.tensor(g, device=dev)
tlab, ttl, tll = (torch.tensor(a, device=dev) for a in (labels, tl, ll))
df, dg, costs = torch.empty_like(tf), torch.empty_like(tg), torch.empty(N, device=dev)
ws = torch.empty(_lib.workspace_bytes_add(T, U, N), dtype=torch.uint8, device=dev)
lib = _lib.lib()
def call():
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream().cuda_stream, blank_label=blank, maxT=T, maxU=U, batch_first=True)
    assert lib.compute_rnnt_loss_add(tf.data_ptr(), tg.data_ptr(), df.data_ptr(), dg.data_ptr(), tlab.data_ptr(), tll.data_ptr(), ttl.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt) == 0
def check(tag, scale):
    z = (f * scale)[:, :, None, :].astype(np.float64) + (g * scale)[:, None, :, :].astype(np.float64)
    ref_c, ref_gz = oracle.rnnt_logits(z, labels, tl, ll, blank)
    edf = np.abs(df.cpu().numpy() - ref_gz.sum(axis=2)); edg = np.abs(dg.cpu().numpy() - ref_gz.sum(axis=1))
    print(tag, scale, "cost err", np.abs(costs.cpu().numpy() - ref_c).max(), "df err", edf.max(), "dg err", edg.max(), "at", np.unravel_index(edg.argmax(), edg.shape), "blank", blank, "labels", labels.tolist(), "ll", ll.tolist())
for rep in range(3):
    call(); torch.cuda.synchronize(); check("eager%d" % rep, 1.0)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): call()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph): call()
for scale in (1.0, 0.5, 1.0):
    tf.copy_(torch.tensor(f * scale)); tg.copy_(torch.tensor(g * scale))
    costs.zero_(); df.zero_(); dg.zero_()
    graph.replay(); torch.cuda.synchronize(); check("graph", scale)
