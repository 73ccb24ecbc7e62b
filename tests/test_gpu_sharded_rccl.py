"""-m gpu, >= 2 devices: the batch-sharded path with TWO ranks over RCCL (backend "nccl" on ROCm), one process per
GPU -- the twin of tests/test_sharded_gloo.py on the real transport.  Every rank runs the HIP hot path on its slab
(compute_rnnt_loss_fwd / _bwd), ONE all-reduce (or all-gather for 'none') carries the loss; losses and gradients
must equal the unsharded single-GPU run.  Skipped on a 1-GPU box (RCCL refuses two ranks on one device).
Also: `bench.py --gpus 2` launches its own two ranks and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(dev):
    g = torch.Generator().manual_seed(11)
    N, T, U, A = 6, 23, 9, 130
    acts = torch.randn(N, T, U, A, generator=g)
    labels = torch.randint(1, A, (N, U - 1), generator=g, dtype=torch.int32)
    tl = torch.tensor([T, 9, T, T, 14, T], dtype=torch.int32)      # every shard contains the maxima
    ll = torch.tensor([U - 1, 3, 0, U - 1, 5, 2], dtype=torch.int32)
    return [t.to(dev) for t in (acts, labels, tl, ll)]


def _worker(rank, world, port, reduction, q, backend="nccl"):
    for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:                                   # both ranks on device 0, collectives over gloo (staged through the host)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from warprnnt_pytorch.sharded import ShardedRNNTLoss
    acts, labels, tl, ll = _batch(dev)
    n = acts.shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    x = acts[sl].clone().requires_grad_(True)
    loss = ShardedRNNTLoss(blank=0, reduction=reduction)(x, labels[sl].contiguous(), tl[sl].contiguous(),
                                                        ll[sl].contiguous())
    w = torch.arange(1, loss.numel() + 1, dtype=loss.dtype, device=dev)
    (loss * w).sum().backward()
    torch.cuda.synchronize()
    q.put((rank, loss.detach().cpu().numpy(), x.grad.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_two_rccl_ranks_equal_single_gpu(reduction):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one RCCL rank per device)")
    import torch.multiprocessing as mp
    from warprnnt_pytorch import RNNTLoss
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    x = acts.clone().requires_grad_(True)
    ref = RNNTLoss(reduction=reduction)(x, labels, tl, ll)
    w = torch.arange(1, ref.numel() + 1, dtype=ref.dtype, device=dev)
    (ref * w).sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduction, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = acts.shape[0] // 2
    for rank, loss, grad in got:
        assert np.allclose(loss, ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)     # global loss on every rank
        assert np.allclose(grad, x.grad[rank * n:(rank + 1) * n].cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_two_ranks_sharing_one_gpu_over_gloo(reduction):
    """What a 1-GPU box can exercise of the sharded DEVICE path at world_size 2: two processes, both on cuda:0, each
    running the HIP hot path (compute_rnnt_loss_fwd / _bwd) on its shard of the batch, the one collective over gloo
    on the device tensors (RCCL refuses two ranks on one device).  Losses and gradients equal the unsharded run."""
    import torch.multiprocessing as mp
    from warprnnt_pytorch import RNNTLoss
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    x = acts.clone().requires_grad_(True)
    ref = RNNTLoss(reduction=reduction)(x, labels, tl, ll)
    w = torch.arange(1, ref.numel() + 1, dtype=ref.dtype, device=dev)
    (ref * w).sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduction, q, "gloo")) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = acts.shape[0] // 2
    for rank, loss, grad in got:
        assert np.allclose(loss, ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert np.allclose(grad, x.grad[rank * n:(rank + 1) * n].cpu().numpy(), atol=1e-6)


def test_bench_self_launches_two_ranks():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    # BASELINE config 5 at every N: the FIXED global batch of 1024 samples, 512 per GPU here
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["config"]["per_gpu_batch"] == 512
    assert line["dtype"] == "bf16" and line["scaling"] == "strong"
    m = line["multi_gpu"]
    assert m["ranks_seen"] == 2 and m["backend"] == "nccl" and len(m["per_rank_ms"]) == 2 and m["devices"] == [0, 1]
    assert m["one_gpu_full_batch_ms"] > 0 and m["speedup"] > 1.0 and m["reduced_loss_agrees"]


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2 --oversubscribe-gloo` on whatever devices there are (a one-GPU box: both ranks on cuda:0, gloo
    instead of RCCL, which refuses two ranks on one device): the self-launcher, the shard sizes of the fixed global
    batch, rank 0's whole-batch run, the per-rank gather and the rank-0-only JSON line all run with world_size 2."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe-gloo", "--global-batch", "64",
                          "--steps", "4", "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[:600]          # ONE line, from rank 0 only
    line = json.loads(out.stdout.strip())
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["dtype"] == "bf16"
    assert line["config"]["global_batch"] == 64 and line["config"]["per_gpu_batch"] == 32 and "c5" in line["config"]["workload"]
    m = line["multi_gpu"]
    assert m["ranks_seen"] == 2 and m["backend"] == "gloo" and len(m["per_rank_ms"]) == 2 and len(m["per_rank_single_gpu_ms"]) == 2
    assert "oversubscribed" in m and m["reduced_loss_agrees"]
    assert m["one_gpu_full_batch_ms"] > 0 and m["speedup"] > 0 and abs(m["strong_scaling_efficiency"] - m["speedup"] / 2) < 1e-3
    assert abs(max(m["per_rank_ms"]) - line["value"]) <= 1e-3 * line["value"]
    assert line["check"]["passed"], line["check"]
    # the one-GPU base of the strong scaling under the SAME key the one-GPU line uses (tests/test_bench_cli.py)
    assert line["other_workloads"]["c5_full_1024_on_one_gpu"]["ms_per_step"] == m["one_gpu_full_batch_ms"]
    # the weak form is still there
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe-gloo", "--per-gpu-batch", "16",
                          "--steps", "2", "--warmup", "1", "--no-verify"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip())
    assert line["scaling"] == "weak" and line["config"]["global_batch"] == 32 and "one_gpu_full_batch_ms" not in line["multi_gpu"]


def test_bench_sharded_step_on_one_rank():
    """The step `bench.py --gpus N` times for N > 1 (loss + gradients -> RCCL all-reduce of [loss sum, count]; and its
    two-phase variant with the collective beside the gradient pass), on a one-rank RCCL group so that it runs on a one-GPU
    box too: same loss sum as the plain one-GPU step on the same inputs, the JSON line as the ONLY line of stdout."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c5", "--steps", "4", "--warmup", "1",
            "--no-cpu-baseline", "--no-traffic-pass"]
    lines = []
    for extra in (["--force-sharded"], ["--force-sharded", "--overlap-collective"], [], ["--force-sharded", "--torch-collective"]):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        assert len(out.stdout.strip().splitlines()) == 1, out.stdout[:400]     # ONE line: RCCL's banner goes to stderr
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
    sh, ov, plain, tc = lines
    # the default sharded step is the library's own (compute_rnnt_loss_sharded); --torch-collective is the round-2 form
    assert "compute_rnnt_loss_sharded" in sh["multi_gpu"]["collective"] and "torch.distributed" in tc["multi_gpu"]["collective"]
    assert abs(tc["check"]["loss_sum"] - plain["check"]["loss_sum"]) <= 1e-6 * abs(plain["check"]["loss_sum"])
    assert abs(ov["check"]["loss_sum"] - plain["check"]["loss_sum"]) <= 1e-6 * abs(plain["check"]["loss_sum"])
    assert ov["stage_ms"]["grad"] > 0 and ov["stage_ms"]["row_stats"] > 0      # two calls, one profiled step
    assert sh["n_gpus"] == 1 and sh["scaling"] == "weak" and sh["dtype"] == "bf16" and sh["value"] > 0
    assert "all-reduce" in sh["config"]["parallelism"]
    assert abs(sh["check"]["loss_sum"] - plain["check"]["loss_sum"]) <= 1e-6 * abs(plain["check"]["loss_sum"])
    assert sh["stage_ms"]["grad"] > 0 and sh["stage_ms"]["row_stats"] > 0
    # the multi-GPU line is self-sufficient: ranks RCCL saw, per-rank times, the same workload without the collective
    for line in (sh, ov, tc):
        m = line["multi_gpu"]
        assert m["ranks_seen"] == 1 and m["backend"] == "nccl" and len(m["per_rank_ms"]) == 1
        assert m["single_gpu_same_workload_ms"] > 0 and 0.3 < m["scaling_efficiency"] <= 1.5
        assert abs(m["per_rank_ms"][0] - line["value"]) <= 1e-3 * line["value"]
    assert "multi_gpu" not in plain
    # every line carries its own parity evidence (two samples of the timed batch against the fp64 oracle)
    for line in lines:
        assert line["check"]["passed"] and line["check"]["max_abs_grad_err"] <= 4e-3, line["check"]
        assert line["check"]["max_err_over_quantum"] <= 1.0, line["check"]      # per element: one rounding of the stored bf16


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus N` with fewer than N devices must fail loudly, never fall back to fewer GPUs."""
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 2
    assert "refusing" in out.stderr and not out.stdout.strip()


# ------------------------------------------------------------------------------------------------------------------
# compute_rnnt_loss_sharded: the sharded step behind the C-ABI, with the collective issued by the library itself
import ctypes as C


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]               # rccl.h: NCCL_UNIQUE_ID_BYTES


def _rccl():
    """The RCCL the library itself resolves (dlopen by soname: the process's loaded copy when there is one)."""
    for name in ("librccl.so.1", "librccl.so"):
        try:
            lib = C.CDLL(name)
        except OSError:
            continue
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        return lib
    pytest.skip("no librccl in this environment")


def _sharded_call(acts, labels, tl, ll, comm):
    """One compute_rnnt_loss_sharded call; returns (costs, grads, [sum, count])."""
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    dev = acts.device
    N, T, U, A = acts.shape
    costs = torch.zeros(N, device=dev)
    grads = torch.full_like(acts, 5.0)
    pair = torch.full((2,), -1.0, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=0, maxT=T,
                           maxU=U, batch_first=True)
    argv = (acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N,
            costs.data_ptr(), None, pair.data_ptr(), comm, ws.data_ptr(), opt, _lib.DT_F32)
    if comm is not None:
        # a communicator the library has not been introduced to is refused before anything is enqueued (rnnt.h: revision 5)
        lib.rnnt_sharded_release(comm)
        assert lib.compute_rnnt_loss_sharded(*argv) == _lib.RNNT_STATUS_INVALID_VALUE
        assert lib.rnnt_sharded_prepare(comm) == 0
    st = lib.compute_rnnt_loss_sharded(*argv)
    assert st == 0, _lib.status_string(st)
    torch.cuda.synchronize(dev)
    if comm is not None:
        lib.rnnt_sharded_release(comm)
    return costs, grads, pair


def test_which_rccl_the_library_calls():
    """compute_rnnt_loss_sharded must call the ncclAllReduce of the RCCL copy that made the communicator.  Unregistered it
    takes the ONE librccl mapped into the process (here: the one PyTorch loaded) and says so; a registered pointer wins;
    NULL forgets it again."""
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    rccl = _rccl()
    lib.rnnt_set_rccl_all_reduce(None)
    src = lib.rnnt_rccl_source().decode()
    assert "librccl" in src, src
    import re
    mapped = {os.path.realpath(m.group(0)) for m in re.finditer(r"/\S*librccl\.so\S*", open("/proc/self/maps").read())}
    assert len(mapped) == 1 and os.path.realpath(src) in mapped, (src, mapped)      # the copy this process already holds
    lib.rnnt_set_rccl_all_reduce(C.cast(rccl.ncclAllReduce, C.c_void_p))
    assert lib.rnnt_rccl_source() == b"registered by the caller"
    lib.rnnt_set_rccl_all_reduce(None)
    assert lib.rnnt_rccl_source().decode() == src


def test_native_sharded_entry_on_one_rank():
    """compute_rnnt_loss_sharded without a communicator (the local [sum, count] pair) and over a ONE-rank RCCL communicator
    created here (a 1-GPU box can run that): costs and gradients of compute_rnnt_loss_async, pair = [sum of the costs in
    fp64, N] either way; argument validation."""
    from warprnnt_pytorch import _lib, warp_rnnt
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    ref_c = torch.zeros(acts.shape[0], device=dev)
    ref_g = torch.zeros_like(acts)
    warp_rnnt.gpu_rnnt_async(acts, labels, tl, ll, ref_c, ref_g, 0)
    torch.cuda.synchronize()
    costs, grads, pair = _sharded_call(acts, labels, tl, ll, None)
    assert torch.equal(costs, ref_c) and torch.equal(grads, ref_g)
    assert pair[1].item() == acts.shape[0] and abs(pair[0].item() - ref_c.double().sum().item()) < 1e-9
    rccl = _rccl()
    uid = _NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        costs2, grads2, pair2 = _sharded_call(acts, labels, tl, ll, comm)
    finally:
        rccl.ncclCommDestroy(comm)
    assert torch.equal(costs2, ref_c) and torch.equal(grads2, ref_g) and torch.equal(pair2, pair)
    lib = _lib.lib()
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, maxT=acts.shape[1], maxU=acts.shape[2], batch_first=True)
    assert lib.compute_rnnt_loss_sharded(acts.data_ptr(), None, labels.data_ptr(), ll.data_ptr(), tl.data_ptr(), acts.shape[3],
                                         acts.shape[0], costs.data_ptr(), None, None, None, costs.data_ptr(), opt, 0) == 2
    # ALL RANKS OR NONE: a rank whose own part fails (here: a blank label outside the vocabulary) still joins the collective,
    # with a NaN pair, and returns its own status -- its peers are not left waiting in ncclAllReduce
    uid = _NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        bad = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=10 ** 6,
                               maxT=acts.shape[1], maxU=acts.shape[2], batch_first=True)
        pair3 = torch.zeros(2, dtype=torch.float64, device=dev)
        assert lib.rnnt_sharded_prepare(comm) == 0
        ws = torch.empty(_lib.workspace_bytes(acts.shape[1], acts.shape[2], acts.shape[0], True, 4), dtype=torch.uint8, device=dev)
        st = lib.compute_rnnt_loss_sharded(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), ll.data_ptr(), tl.data_ptr(),
                                           acts.shape[3], acts.shape[0], costs.data_ptr(), None, pair3.data_ptr(), comm,
                                           ws.data_ptr(), bad, 0)
        torch.cuda.synchronize(dev)
        assert st == 2 and torch.isnan(pair3).all()
    finally:
        lib.rnnt_sharded_release(comm)
        rccl.ncclCommDestroy(comm)


def _native_worker(rank, world, uid_bytes, q):
    for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    rccl = _rccl()
    uid = _NcclUniqueId()
    C.memmove(C.byref(uid), uid_bytes, 128)
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    acts, labels, tl, ll = _batch(dev)
    sl = slice(0, 4) if rank == 0 else slice(4, 6)                  # ragged shards: 4 + 2 samples
    costs, grads, pair = _sharded_call(acts[sl].contiguous(), labels[sl].contiguous(), tl[sl].contiguous(), ll[sl].contiguous(), comm)
    q.put((rank, costs.cpu().numpy(), pair.cpu().numpy()))
    rccl.ncclCommDestroy(comm)


def test_native_sharded_entry_over_two_rccl_ranks():
    """Two processes, one GPU each, ragged shards (4 + 2 samples): every rank's pair is the GLOBAL [sum, count]."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one RCCL rank per device)")
    import torch.multiprocessing as mp
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    ref_c = torch.zeros(acts.shape[0], device=dev)
    warp_rnnt.gpu_rnnt_async(acts, labels, tl, ll, ref_c, torch.zeros_like(acts), 0)
    torch.cuda.synchronize()
    rccl = _rccl()
    uid = _NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid_bytes = C.string_at(C.addressof(uid), 128)                 # all 128 bytes (a c_char field reads as a NUL-terminated string)
    procs = [ctx.Process(target=_native_worker, args=(r, 2, uid_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = ref_c.double().sum().item()
    for rank, costs, pair in got:
        assert pair[1] == 6 and abs(pair[0] - total) <= 1e-9 * abs(total)
        assert np.array_equal(costs, ref_c[:4].cpu().numpy() if rank == 0 else ref_c[4:].cpu().numpy())


def _gloo_carried_worker(rank, world, port, fail_rank, q):
    """One rank of a two-rank job whose ranks SHARE device 0: compute_rnnt_loss_sharded with the collective carried by gloo
    through a registered all-reduce function (the library calls it exactly where it would call ncclAllReduce)."""
    for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from warprnnt_pytorch import _lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _lib.lib()
    acts, labels, tl, ll = _batch(dev)
    sl = slice(0, 4) if rank == 0 else slice(4, 6)                  # ragged shards: 4 + 2 samples
    a, lab, t, l = (v[sl].contiguous() for v in (acts, labels, tl, ll))
    N, T, U, A = a.shape
    costs, grads = torch.zeros(N, device=dev), torch.zeros_like(a)
    pair = torch.full((2,), -1.0, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.workspace_bytes(T, U, N, True, 4), dtype=torch.uint8, device=dev)
    calls = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    def all_reduce(send, recv, count, dtype, op, comm, stream):
        # ncclAllReduce's signature; in place on `pair` (the library passes its loss_sum_count_device twice)
        calls.append((send == pair.data_ptr() and recv == pair.data_ptr(), count, dtype, op, comm))
        torch.cuda.synchronize(dev)
        host = pair.cpu()
        dist.all_reduce(host)
        pair.copy_(host)
        return 0

    token = C.c_void_p(0xC0FFEE0 + rank)                             # stands for an ncclComm_t
    lib.rnnt_set_rccl_all_reduce(C.cast(all_reduce, C.c_void_p))
    assert lib.rnnt_sharded_prepare(token) == 0
    # a shard only THIS rank cannot run: its blank label lies outside the vocabulary
    blank = 10 ** 6 if rank == fail_rank else 0
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=blank, maxT=T,
                           maxU=U, batch_first=True)
    st = lib.compute_rnnt_loss_sharded(a.data_ptr(), grads.data_ptr(), lab.data_ptr(), l.data_ptr(), t.data_ptr(), A, N,
                                       costs.data_ptr(), None, pair.data_ptr(), token, ws.data_ptr(), opt, _lib.DT_F32)
    torch.cuda.synchronize(dev)
    q.put((rank, st, pair.cpu().numpy(), costs.cpu().numpy(), calls))
    lib.rnnt_sharded_release(token)
    lib.rnnt_set_rccl_all_reduce(None)
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1, 0])
def test_two_ranks_all_or_none_over_a_registered_all_reduce(fail_rank):
    """The native sharded step with world = 2: both ranks on device 0, the 16-byte all-reduce
    carried by gloo through rnnt_set_rccl_all_reduce.  Healthy: every rank's pair is the GLOBAL [sum, count] of 4 + 2 samples.
    One rank's shard fails: that rank STILL joins the collective, with a NaN pair, so every rank's reduced pair is NaN and
    nobody is left blocked; it returns its own status, its peer SUCCESS."""
    import socket
    import torch.multiprocessing as mp
    from warprnnt_pytorch import warp_rnnt
    dev = torch.device("cuda:0")
    acts, labels, tl, ll = _batch(dev)
    ref_c = torch.zeros(acts.shape[0], device=dev)
    warp_rnnt.gpu_rnnt_async(acts, labels, tl, ll, ref_c, torch.zeros_like(acts), 0)
    torch.cuda.synchronize()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_carried_worker, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = ref_c.double().sum().item()
    for rank, st, pair, costs, calls in got:
        assert len(calls) == 1 and calls[0][0] and calls[0][1:4] == (2, 8, 0)        # in place, 2 x ncclFloat64, ncclSum
        if fail_rank < 0:
            assert st == 0 and pair[1] == 6 and abs(pair[0] - total) <= 1e-9 * abs(total)
            assert np.array_equal(costs, ref_c[:4].cpu().numpy() if rank == 0 else ref_c[4:].cpu().numpy())
        else:
            assert st == (2 if rank == fail_rank else 0)
            assert np.isnan(pair).all()                                              # every rank sees that the step failed


def _duplicate_device_worker(rank, world, uid_bytes, q):
    """The body of _native_worker up to the communicator, with BOTH ranks on device 0."""
    for p in (ROOT, os.path.join(ROOT, "warp-transducer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    acts, labels, tl, ll = _batch(dev)                               # the shard this rank would have run
    rccl = _rccl()
    uid = _NcclUniqueId()
    C.memmove(C.byref(uid), uid_bytes, 128)
    comm = C.c_void_p()
    rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc == 0:                                                      # (an RCCL that accepts it: run the real thing, then)
        sl = slice(0, 4) if rank == 0 else slice(4, 6)
        costs, grads, pair = _sharded_call(acts[sl].contiguous(), labels[sl].contiguous(), tl[sl].contiguous(), ll[sl].contiguous(), comm)
        q.put((rank, rc, pair.cpu().numpy()))
        rccl.ncclCommDestroy(comm)
    else:
        q.put((rank, rc, None))


def test_two_native_ranks_on_one_device_reach_rccl_and_are_refused():
    """What a ONE-GPU box can execute of the two-rank native path: both processes go through the
    whole bootstrap -- unique id carried between the processes, device selection, ncclCommInitRank with world = 2 -- on device
    0, where RCCL must answer with its duplicate-GPU refusal (ncclInvalidUsage = 5) on both ranks rather than hang or crash.
    (If an RCCL build ever accepts two ranks on one device, the sharded call itself runs and must give the global pair.)"""
    import queue as queue_mod
    import torch.multiprocessing as mp
    rccl = _rccl()
    uid = _NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid_bytes = C.string_at(C.addressof(uid), 128)
    procs = [ctx.Process(target=_duplicate_device_worker, args=(r, 2, uid_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in procs:
            got.append(q.get(timeout=150))
    except queue_mod.Empty:
        pass
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    assert len(got) == 2, "a rank never came back from ncclCommInitRank (world 2 on one device)"
    codes = sorted(rc for _, rc, _ in got)
    if codes == [0, 0]:
        assert all(pair[1] == 6 for _, _, pair in got)
    else:
        assert all(rc != 0 for rc in codes), codes                  # refused on BOTH ranks (5 = ncclInvalidUsage: duplicate GPU)
