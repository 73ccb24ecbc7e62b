"""RNN-T loss on PACKED ("compact") joint activations: sample b contributes only its T_b x U_b real rows.

The reference pads every sample of the joint tensor to (max T, max U+1) and writes zeros over the padding
of the gradient (pytorch_binding/warprnnt_pytorch/__init__.py:24, include/detail/gpu_rnnt_kernel.h:159).
With the lengths of a real batch spread over [max/2, max] that padding is ~40 % of the tensor.  Here

    acts : (sum_b T_b * (U_b + 1), V)       row (b, t, u) = offsets[b] + t * (U_b + 1) + u

(`pack_joint` builds it from a padded tensor; a joint network evaluated on gathered (t, u) pairs produces it
directly), and the library (compute_rnnt_loss_packed_fwd / _bwd of include/rnnt.h) neither reads nor writes a
byte of padding.  Same values as ``RNNTLoss`` on the padded tensor.  CPU tensors take the library's RNNT_CPU
location (log_softmax applied here, gradients computed in forward), as `RNNTLoss` does.
"""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import _lib, check_contiguous, check_dim, check_type
from .warp_rnnt import _DT

__all__ = ["rnnt_loss_packed", "RNNTLossPacked", "pack_joint", "unpack_joint", "row_offsets"]


def row_offsets(act_lens, label_lens):
    """int64 (N+1,) cumulative row counts: offsets[b+1] - offsets[b] = T_b * (U_b + 1)."""
    rows = act_lens.to(torch.int64) * (label_lens.to(torch.int64) + 1)
    return torch.cat([rows.new_zeros(1), rows.cumsum(0)])


def pack_joint(acts, act_lens, label_lens):
    """(N, T, U+1, V) padded -> (sum T_b (U_b+1), V) packed."""
    return torch.cat([acts[b, :int(t), :int(l) + 1].reshape(-1, acts.shape[-1])
                      for b, (t, l) in enumerate(zip(act_lens.tolist(), label_lens.tolist()))])


def unpack_joint(packed, act_lens, label_lens, T=None, U=None):
    """Inverse of `pack_joint`; the padding is zero-filled."""
    tl, ll = act_lens.tolist(), label_lens.tolist()
    T = max(tl) if T is None else T
    U = max(ll) + 1 if U is None else U
    out = packed.new_zeros((len(tl), T, U, packed.shape[-1]))
    at = 0
    for b, (t, l) in enumerate(zip(tl, ll)):
        n = t * (l + 1)
        out[b, :t, :l + 1] = packed[at:at + n].reshape(t, l + 1, -1)
        at += n
    return out


class _RNNTPacked(Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, max_T, max_U):
        check_type(labels, torch.int32, "labels")
        check_type(label_lens, torch.int32, "label_lengths")
        check_type(act_lens, torch.int32, "lengths")
        for var, name in ((acts, "acts"), (labels, "labels"), (act_lens, "lengths"), (label_lens, "label_lengths")):
            check_contiguous(var, name)
        check_dim(acts, 2, "acts")
        check_dim(labels, 2, "labels")
        check_dim(act_lens, 1, "lengths")
        check_dim(label_lens, 1, "label_lengths")
        if acts.dtype not in _DT:
            raise TypeError("unsupported dtype %s" % acts.dtype)
        N = act_lens.shape[0]
        if label_lens.shape[0] != N or labels.shape[0] != N:
            raise ValueError("must have a length per example.")
        offs = row_offsets(act_lens, label_lens)
        ctx.on_gpu = acts.is_cuda
        if not acts.is_cuda:
            return _RNNTPacked._forward_cpu(ctx, acts, labels, act_lens, label_lens, offs, blank, reduction,
                                            fastemit_lambda, max_T, max_U)
        lazy_check = not (max_T is None or max_U is None)
        if max_T is None or max_U is None:
            # one host round trip for the lattice dimensions (pass max_T / max_U to avoid it)
            mt, ml, total = torch.stack([act_lens.max().to(torch.int64), label_lens.max().to(torch.int64),
                                         offs[-1]]).tolist()
            max_T, max_U = mt, ml + 1
            if total != acts.shape[0]:
                raise ValueError("acts has %d rows, the lengths describe %d" % (acts.shape[0], total))
        if labels.shape[1] != max_U - 1:
            raise ValueError("Output length mismatch")
        lib = _lib.lib()
        R, V = acts.shape
        code, esz = _DT[acts.dtype]
        dev = acts.device
        cost_dtype = torch.float64 if acts.dtype == torch.float64 else torch.float32
        need_grad = acts.requires_grad
        with torch.cuda.device(dev):
            costs = torch.empty(N, dtype=cost_dtype, device=dev)
            ws = torch.empty(_lib.workspace_bytes(max_T, max_U, N, True, esz), dtype=torch.uint8, device=dev)
            opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0,
                                   stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=int(blank),
                                   maxT=int(max_T), maxU=int(max_U), batch_first=True)
            lab_ptr = labels.data_ptr() if labels.numel() else costs.data_ptr()   # maxU == 1: never read
            st = lib.compute_rnnt_loss_packed_fwd(acts.data_ptr(), lab_ptr, label_lens.data_ptr(), act_lens.data_ptr(),
                                                  offs.data_ptr(), R, V, N, costs.data_ptr(), ws.data_ptr(), opt, code,
                                                  1 if need_grad else 0, float(fastemit_lambda))
            _lib.check(st, "compute_rnnt_loss_packed_fwd")
            ws.record_stream(torch.cuda.current_stream(dev))
            if lazy_check:
                # max_T / max_U were supplied, so nothing was brought to the host: the consistency of the row count is
                # checked on the device and a mismatch poisons the loss (too-small maxima poison the affected samples
                # inside the library; nothing is read or written out of bounds either way)
                costs = torch.where(offs[-1] == R, costs, torch.full_like(costs, float("nan")))
        ctx.save_for_backward(acts, offs)
        ctx.workspace = ws if need_grad else None
        ctx.opt_dims = (int(blank), int(max_T), int(max_U), N)
        ctx.mean_scale = 1.0 / N if reduction == "mean" else 1.0
        if reduction in ("sum", "mean"):
            costs = costs.sum().unsqueeze_(-1)
            if reduction == "mean":
                costs /= N
        return costs

    @staticmethod
    def _forward_cpu(ctx, logp, labels, act_lens, label_lens, offs, blank, reduction, fastemit_lambda, max_T, max_U):
        """RNNT_CPU location (compute_rnnt_loss_packed with host arrays): `logp` are LOG-PROBABILITIES (the caller,
        `rnnt_loss_packed`, applied log_softmax as `rnnt_loss` does on the CPU); gradients are computed here and
        scaled in backward, as in the reference's flow."""
        if logp.dtype not in (torch.float32, torch.float64):
            raise TypeError("the CPU location takes float32 or float64")
        if fastemit_lambda:
            raise NotImplementedError("fastemit_lambda is an extension of the GPU route")
        N = act_lens.shape[0]
        max_T = int(act_lens.max()) if max_T is None else int(max_T)
        max_U = int(label_lens.max()) + 1 if max_U is None else int(max_U)
        if int(offs[-1]) != logp.shape[0]:
            raise ValueError("acts has %d rows, the lengths describe %d" % (logp.shape[0], int(offs[-1])))
        if labels.shape[1] != max_U - 1:
            raise ValueError("Output length mismatch")
        lib = _lib.lib()
        R, V = logp.shape
        code, esz = _DT[logp.dtype]
        costs = torch.zeros(N, dtype=logp.dtype)
        grads = torch.empty_like(logp) if logp.requires_grad else None
        ws = torch.empty(_lib.workspace_bytes(max_T, max_U, N, False, esz), dtype=torch.uint8)
        opt = _lib.rnntOptions(loc=_lib.RNNT_CPU, num_threads=0, stream=None, blank_label=int(blank), maxT=max_T,
                               maxU=max_U, batch_first=True)
        lab_ptr = labels.data_ptr() if labels.numel() else costs.data_ptr()
        st = lib.compute_rnnt_loss_packed(logp.data_ptr(), grads.data_ptr() if grads is not None else None, lab_ptr,
                                          label_lens.data_ptr(), act_lens.data_ptr(), offs.data_ptr(), R, V, N,
                                          costs.data_ptr(), None, ws.data_ptr(), opt, code, 0.0)
        _lib.check(st, "compute_rnnt_loss_packed")
        ctx.cpu_grads = grads
        ctx.cpu_rows = offs[1:] - offs[:-1]
        ctx.mean_scale = 1.0 / N if reduction == "mean" else 1.0
        if reduction in ("sum", "mean"):
            costs = costs.sum().unsqueeze_(-1)
            if reduction == "mean":
                costs /= N
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        if not ctx.on_gpu:
            n = ctx.cpu_rows.shape[0]
            per_sample = (grad_output.reshape(-1).to(ctx.cpu_grads.dtype) * ctx.mean_scale).expand(n)
            per_row = torch.repeat_interleave(per_sample, ctx.cpu_rows)
            return ctx.cpu_grads * per_row.unsqueeze(1), None, None, None, None, None, None, None, None
        acts, offs = ctx.saved_tensors
        blank, max_T, max_U, N = ctx.opt_dims
        lib = _lib.lib()
        R, V = acts.shape
        code, _ = _DT[acts.dtype]
        dev = acts.device
        sdt = torch.float64 if acts.dtype == torch.float64 else torch.float32
        with torch.cuda.device(dev):
            scale = (grad_output.reshape(-1).to(device=dev, dtype=sdt) * ctx.mean_scale).expand(N).contiguous()
            grads = torch.empty_like(acts)
            opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0,
                                   stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=blank,
                                   maxT=max_T, maxU=max_U, batch_first=True)
            st = lib.compute_rnnt_loss_packed_bwd(acts.data_ptr(), grads.data_ptr(), scale.data_ptr(), offs.data_ptr(),
                                                  R, V, N, ctx.workspace.data_ptr(), opt, code)
            _lib.check(st, "compute_rnnt_loss_packed_bwd")
            ctx.workspace.record_stream(torch.cuda.current_stream(dev))
        return grads, None, None, None, None, None, None, None, None


def rnnt_loss_packed(acts, labels, act_lens, label_lens, blank=0, reduction="mean", fastemit_lambda=0.0,
                     max_T=None, max_U=None):
    """RNN-T loss of packed activations ``acts`` (sum_b T_b (U_b+1), V); other arguments as `rnnt_loss`.
    ``max_T`` / ``max_U`` (= max label length + 1): the batch maxima, if the caller knows them on the host."""
    if not acts.is_cuda:
        acts = torch.nn.functional.log_softmax(acts, -1)      # the CPU location takes log-probabilities
    return _RNNTPacked.apply(acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, max_T, max_U)


class RNNTLossPacked(Module):
    def __init__(self, blank=0, reduction="mean", fastemit_lambda=0.0):
        super().__init__()
        self.blank, self.reduction, self.fastemit_lambda = blank, reduction, fastemit_lambda

    def forward(self, acts, labels, act_lens, label_lens, max_T=None, max_U=None):
        return rnnt_loss_packed(acts, labels, act_lens, label_lens, self.blank, self.reduction,
                                self.fastemit_lambda, max_T, max_U)
