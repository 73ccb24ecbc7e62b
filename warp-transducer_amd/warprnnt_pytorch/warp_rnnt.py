"""`warprnnt_pytorch.warp_rnnt` -- the extension-module surface of the reference binding.

Same two entry points and the same 8-argument signature as the pybind11 module of
pytorch_binding/src/binding.cpp:12-19,84-91,157-162:

    cpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, blank_label, num_threads)
    gpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, blank_label, num_threads)

They read (N,T,U,A) off ``acts`` (binding.cpp:26-30,93-96), size and allocate a temporary
workspace from the framework allocator (binding.cpp:115-120), use the current stream
(binding.cpp:104) and dispatch on dtype (binding.cpp:46-81,111-153) -- float32 and float64
as the reference, plus bfloat16/float16 on the GPU.  An empty ``grads`` tensor means
"score only" (reference __init__.py:24 passes torch.zeros(0)).
"""
import os as _os

import torch

from . import _lib
from ._checks import check_gpu_arguments

# The compiled module (csrc/binding.cpp, built in tree by build_ext.py / setup.py) is the default binding, as in the
# reference (a pybind11 extension, pytorch_binding/src/binding.cpp:157-162); the ctypes route below is the fallback when it
# has not been built, and WARPRNNT_BINDING=ctypes | ext selects one explicitly ("ext" fails loudly if it is missing).
# Either way every call ends in the C-ABI of libwarprnnt.so: there is no Python or CPU fallback for the HIP path.
# The compiled module is LINKED to the libwarprnnt.so it was built with (in tree: ../lib, installed: ./lib); a WARP_RNNT_PATH that
# names another library (development builds: lib/dev) can only be honoured by the ctypes loader, so it selects that one -- two
# copies of the library in one process would each have their own thread-local and profiling state.
_EXT = None
_want = _os.environ.get("WARPRNNT_BINDING", "auto").lower()
if _want == "auto" and _os.environ.get("WARP_RNNT_PATH") and \
        _os.path.realpath(_lib.library_path()) not in (_os.path.realpath(_os.path.join(_lib._HERE, "lib", "libwarprnnt.so")),
                                                       _os.path.realpath(_os.path.join(_lib._HERE, "..", "lib", "libwarprnnt.so"))):
    _want = "ctypes"
if _want != "ctypes":
    try:
        from . import _warp_rnnt_ext as _EXT
    except ImportError:
        if _want == "ext":
            raise
        _EXT = None


_DT = {torch.float32: (_lib.DT_F32, 4), torch.float64: (_lib.DT_F64, 8),
       torch.bfloat16: (_lib.DT_BF16, 2), torch.float16: (_lib.DT_F16, 2)}


def set_aux_stream(stream):
    """Hand the library a second stream (a torch.cuda.Stream of the tensors' device, or None to take it back) for the CALLING
    thread: on long lattices (768 anti-diagonals and more, two samples and more) the one-call entries then run the lattice
    kernel of one half of the batch on it beside the streaming kernels of the other half (include/rnnt.h: rnnt_set_aux_stream;
    same bits; measured on N=64,T=1500,U=301,A=50: 3.4-3.6 -> 3.3-3.5 ms through compute_rnnt_loss, but SLOWER through the
    two-phase pair the autograd wrappers use -- 3.60 -> 3.95 ms -- which is why no wrapper sets it on its own)."""
    _lib.lib().rnnt_set_aux_stream(None if stream is None else stream.cuda_stream)


def binding():
    """'ext' (the compiled PyTorch extension module) or 'ctypes'."""
    return "ext" if _EXT is not None else "ctypes"


def _options(loc, acts, blank_label, num_threads, stream):
    return _lib.rnntOptions(loc=loc, num_threads=max(int(num_threads), 0), stream=stream,
                            blank_label=int(blank_label), maxT=acts.size(1), maxU=acts.size(2),
                            batch_first=True)


def _ptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


def cpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, blank_label, num_threads):
    """RNNT_CPU: acts are LOG-PROBS, grads the sparse d/d(log-probs).  Returns 0 / -1."""
    if _EXT is not None and acts.dtype in (torch.float32, torch.float64):    # (an unsupported dtype: the line on stderr and -1 below)
        return _EXT.cpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, int(blank_label), int(num_threads))
    lib = _lib.lib()
    N, T, U, A = acts.shape
    if acts.dtype == torch.float32:
        fn, esz = lib.compute_rnnt_loss, 4
    elif acts.dtype == torch.float64:
        fn, esz = lib.compute_rnnt_loss_fp64, 8
    else:
        import sys
        print("warp_rnnt.cpu_rnnt: unsupported data type %s" % acts.dtype, file=sys.stderr)
        return -1
    ws = torch.empty(_lib.workspace_bytes(T, U, N, False, esz), dtype=torch.uint8)
    opt = _options(_lib.RNNT_CPU, acts, blank_label, num_threads, None)
    st = fn(acts.data_ptr(), _ptr(grads), labels.data_ptr(), label_lengths.data_ptr(),
            input_lengths.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
    _lib.check(st, "compute_rnnt_loss (RNNT_CPU)")
    return 0


def gpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, blank_label, num_threads, workspace=None):
    """RNNT_GPU: acts are raw LOGITS on an MI355X, grads the dense d/d(logits);
    labels/lengths are device tensors, ``costs`` is a HOST tensor.  Returns 0 / -1.
    (``workspace``: optional caller-owned uint8 device tensor of get_workspace_size bytes; the
    reference's 8-argument form allocates a temporary one from the framework allocator, binding.cpp:120,128.)"""
    if not acts.is_cuda:
        raise ValueError("gpu_rnnt needs device tensors")
    if _EXT is not None and acts.dtype in _DT:
        return _EXT.gpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, int(blank_label), int(num_threads), workspace)
    lib = _lib.lib()
    N, T, U, A = acts.shape
    table = {torch.float32: (lib.compute_rnnt_loss, 4), torch.float64: (lib.compute_rnnt_loss_fp64, 8),
             torch.bfloat16: (lib.compute_rnnt_loss_bf16, 2), torch.float16: (lib.compute_rnnt_loss_fp16, 2)}
    if acts.dtype not in table:
        import sys
        print("warp_rnnt.gpu_rnnt: unsupported data type %s" % acts.dtype, file=sys.stderr)
        return -1
    fn, esz = table[acts.dtype]
    check_gpu_arguments(acts, labels, input_lengths, label_lengths, workspace, _lib.workspace_bytes(T, U, N, True, esz))
    with torch.cuda.device(acts.device):
        ws = workspace if workspace is not None else torch.empty(_lib.workspace_bytes(T, U, N, True, esz),
                                                                 dtype=torch.uint8, device=acts.device)
        stream = torch.cuda.current_stream(acts.device).cuda_stream
        opt = _options(_lib.RNNT_GPU, acts, blank_label, num_threads, stream)
        st = fn(acts.data_ptr(), _ptr(grads), labels.data_ptr(), label_lengths.data_ptr(),
                input_lengths.data_ptr(), A, N, costs.data_ptr(), ws.data_ptr(), opt)
    _lib.check(st, "compute_rnnt_loss (RNNT_GPU)")
    return 0


def gpu_rnnt_async(acts, labels, input_lengths, label_lengths, costs_device, grads, blank_label,
                   grad_scale=None, workspace=None, fastemit_lambda=0.0):
    """Extension: enqueue only (no host copy, no synchronisation).  ``costs_device`` is a device
    tensor (float32, or float64 for float64 acts).  Returns the workspace tensor, which the caller
    must keep alive until the stream has passed this work.  ``fastemit_lambda`` != 0 selects
    compute_rnnt_loss_fastemit."""
    lib = _lib.lib()
    N, T, U, A = acts.shape
    code = {torch.float32: (_lib.DT_F32, 4), torch.float64: (_lib.DT_F64, 8),
            torch.bfloat16: (_lib.DT_BF16, 2), torch.float16: (_lib.DT_F16, 2)}[acts.dtype]
    check_gpu_arguments(acts, labels, input_lengths, label_lengths, workspace, _lib.workspace_bytes(T, U, N, True, code[1]))
    with torch.cuda.device(acts.device):
        if workspace is None:
            workspace = torch.empty(_lib.workspace_bytes(T, U, N, True, code[1]), dtype=torch.uint8,
                                    device=acts.device)
        stream = torch.cuda.current_stream(acts.device).cuda_stream
        opt = _options(_lib.RNNT_GPU, acts, blank_label, 0, stream)
        if fastemit_lambda:
            st = lib.compute_rnnt_loss_fastemit(acts.data_ptr(), _ptr(grads), labels.data_ptr(),
                                                label_lengths.data_ptr(), input_lengths.data_ptr(), A, N,
                                                costs_device.data_ptr(), _ptr(grad_scale), workspace.data_ptr(),
                                                opt, code[0], float(fastemit_lambda))
        else:
            st = lib.compute_rnnt_loss_async(acts.data_ptr(), _ptr(grads), labels.data_ptr(),
                                             label_lengths.data_ptr(), input_lengths.data_ptr(), A, N,
                                             costs_device.data_ptr(), _ptr(grad_scale), workspace.data_ptr(),
                                             opt, code[0])
    _lib.check(st, "compute_rnnt_loss_async")
    return workspace


# Host-side cost of the two-phase entries (what a PyTorch user of SMALL problems sees: on N=16,T=150,U=41,A=28 the
# kernels take ~30 us, tools/autograd_profile.py): the workspace size is queried once per (T, U, N, element size),
# the raw stream handle comes straight from the allocator's stream table, and the device guard is only entered when
# the tensors do not live on the thread's current device.
_WS_BYTES = {}


def _workspace_bytes_cached(T, U, N, esz):
    key = (T, U, N, esz)
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = _lib.workspace_bytes(T, U, N, True, esz)
    return n


# torch._C._cuda_getCurrentRawStream / _cuda_getDevice are private: a torch build without them takes the public route
# (a Stream object per call: a few microseconds more, nothing else changes).
try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _current_device = torch._C._cuda_getDevice
    _raw_stream.__call__, _current_device.__call__
except AttributeError:                                           # pragma: no cover -- depends on the torch build
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

    def _current_device():
        return torch.cuda.current_device()


class _on_device(object):
    """`with torch.cuda.device(d)` only when d is not already the current device (the library launches on the
    calling thread's current HIP device)."""
    __slots__ = ("guard",)

    def __init__(self, index):
        self.guard = None if _current_device() == index else torch.cuda.device(index)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


def gpu_rnnt_fwd(acts, labels, input_lengths, label_lengths, costs_device, blank_label, prepare_backward,
                 fastemit_lambda=0.0):
    """Extension: forward phase only (compute_rnnt_loss_fwd; compute_rnnt_loss_fwd_fastemit when
    ``fastemit_lambda`` is not 0).  Returns the workspace tensor; with ``prepare_backward`` it holds the
    gradient-coefficient table `gpu_rnnt_bwd` needs and must be kept (untouched) until then.  Enqueue only."""
    if _EXT is not None:
        ws = _EXT.gpu_rnnt_fwd(acts, labels, input_lengths, label_lengths, costs_device, int(blank_label), bool(prepare_backward),
                               float(fastemit_lambda))
        ws._rnnt_stream = _raw_stream(acts.device.index)
        return ws
    lib = _lib.lib()
    N, T, U, A = acts.shape
    dt = _DT.get(acts.dtype)
    if dt is None:
        raise TypeError("rnnt_loss: unsupported dtype %s for the GPU location" % acts.dtype)
    code, esz = dt
    check_gpu_arguments(acts, labels, input_lengths, label_lengths)
    index = acts.device.index
    with _on_device(index):
        ws = torch.empty(_workspace_bytes_cached(T, U, N, esz), dtype=torch.uint8, device=acts.device)
        stream = _raw_stream(index)
        # The workspace is allocated on, and used on, the stream that is current HERE: the caching allocator's own
        # bookkeeping covers that stream, so no record_stream.  gpu_rnnt_bwd checks that it runs on the same stream
        # (autograd replays backward on the forward stream) and tells the allocator if it does not.
        ws._rnnt_stream = stream
        opt = _lib.rnntOptions(_lib.RNNT_GPU, 0, stream, int(blank_label), T, U, True)
        lab_ptr = labels.data_ptr() if labels.numel() else costs_device.data_ptr()   # maxU == 1: never read
        if fastemit_lambda:
            st = lib.compute_rnnt_loss_fwd_fastemit(acts.data_ptr(), lab_ptr, label_lengths.data_ptr(),
                                                    input_lengths.data_ptr(), A, N, costs_device.data_ptr(),
                                                    ws.data_ptr(), opt, code, 1 if prepare_backward else 0,
                                                    float(fastemit_lambda))
        else:
            st = lib.compute_rnnt_loss_fwd(acts.data_ptr(), lab_ptr, label_lengths.data_ptr(),
                                           input_lengths.data_ptr(), A, N, costs_device.data_ptr(), ws.data_ptr(), opt,
                                           code, 1 if prepare_backward else 0)
    if st != 0:
        _lib.check(st, "compute_rnnt_loss_fwd")
    return ws


def gpu_rnnt_bwd(acts, grads, grad_scale, workspace, blank_label):
    """Extension: gradient phase (compute_rnnt_loss_bwd) from the workspace of `gpu_rnnt_fwd`;
    ``grad_scale`` is a per-sample device vector (float32; float64 for float64 acts) or None."""
    if _EXT is not None:
        _EXT.gpu_rnnt_bwd(acts, grads, grad_scale, workspace, int(blank_label), int(getattr(workspace, "_rnnt_stream", 0) or 0))
        return 0
    lib = _lib.lib()
    N, T, U, A = acts.shape
    code, esz = _DT[acts.dtype]
    check_gpu_arguments(acts, workspace=workspace, workspace_bytes=_workspace_bytes_cached(T, U, N, esz))
    index = acts.device.index
    with _on_device(index):
        stream = _raw_stream(index)
        if getattr(workspace, "_rnnt_stream", stream) != stream:
            # backward under another torch.cuda.stream than forward (a custom engine, retain_graph replays): the
            # workspace is now in use on a stream its allocation does not know about
            workspace.record_stream(torch.cuda.current_stream(index))
        opt = _lib.rnntOptions(_lib.RNNT_GPU, 0, stream, int(blank_label), T, U, True)
        st = lib.compute_rnnt_loss_bwd(acts.data_ptr(), grads.data_ptr(), _ptr(grad_scale), A, N,
                                       workspace.data_ptr(), opt, code)
    if st != 0:
        _lib.check(st, "compute_rnnt_loss_bwd")
    return 0
