"""Additive-joint ("add network") RNN-T loss: the joint tensor is never materialised.

The reference's add_network branch is called as ``fn(trans_acts, pred_acts, labels, lengths,
label_lengths)`` (pytorch_binding/test/test_time.py:51-70) on the transcription output
``trans_acts`` (B,T,V) and the prediction output ``pred_acts`` (B,U+1,V) of Graves' 2012
transducer, whose joint logits are ``trans_acts[:, :, None] + pred_acts[:, None]``
(docs/rnnt_notes.tex:56-59).  Mathematically this module equals

    RNNTLoss(...)(trans_acts.unsqueeze(2) + pred_acts.unsqueeze(1), labels, act_lens, label_lens)

but it binds ``compute_rnnt_loss_add`` of include/rnnt.h: exp(f+g) = exp(f) exp(g), so the partition
function and d(trans_acts) = sum_u, d(pred_acts) = sum_t (docs/rnnt_notes.tex:147-153) are three
small GEMMs per sample on the fp32 matrix cores (csrc/rnnt_joint_kernels.h).  GPU.  float32, bfloat16 or
float16 activations (storage type; the kernels compute in fp32 and write the gradients in the activations'
dtype -- compute_rnnt_loss_add_fwd_dt / _bwd_dt); the loss is float32.
"""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import _lib, check_contiguous, check_dim, check_type
from ._checks import check_gpu_arguments

__all__ = ["rnnt_loss_add", "RNNTLossAdd"]

_DT = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}


def _certify(trans_acts, pred_acts, labels, act_lens, label_lens, read_lengths=True):
    check_type(labels, torch.int32, "labels")
    check_type(label_lens, torch.int32, "label_lengths")
    check_type(act_lens, torch.int32, "lengths")
    for var, name in ((trans_acts, "trans_acts"), (pred_acts, "pred_acts"), (labels, "labels"),
                      (act_lens, "lengths"), (label_lens, "label_lengths")):
        check_contiguous(var, name)
    check_dim(trans_acts, 3, "trans_acts")
    check_dim(pred_acts, 3, "pred_acts")
    check_dim(labels, 2, "labels")
    check_dim(act_lens, 1, "lengths")
    check_dim(label_lens, 1, "label_lengths")
    if not (trans_acts.is_cuda and pred_acts.is_cuda):
        raise ValueError("the additive-joint loss runs on the GPU only")
    if trans_acts.dtype not in _DT or pred_acts.dtype is not trans_acts.dtype:
        raise TypeError("trans_acts and pred_acts must both be torch.float32, torch.bfloat16 or torch.float16")
    B, T, V = trans_acts.shape
    if pred_acts.shape[0] != B or pred_acts.shape[2] != V:
        raise ValueError("trans_acts (B,T,V) and pred_acts (B,U+1,V) disagree")
    if act_lens.shape[0] != B or label_lens.shape[0] != B:
        raise ValueError("must have a length per example.")
    if pred_acts.device != trans_acts.device:
        raise ValueError("pred_acts must be on the device of trans_acts")
    check_gpu_arguments(trans_acts, labels, act_lens, label_lens)
    if not read_lengths:
        return
    if T != torch.max(act_lens):
        raise ValueError("Input length mismatch")
    if pred_acts.shape[1] != torch.max(label_lens) + 1:
        raise ValueError("Output length mismatch")


class _RNNTAdd(Function):
    """Two-phase (compute_rnnt_loss_add_fwd / _bwd): forward leaves only the workspace behind, backward
    runs the gradient kernels once with grad_output (and 1/B for 'mean') folded in -- no extra torch
    pass over d(trans_acts) / d(pred_acts), as in `warprnnt_pytorch._RNNT`."""

    @staticmethod
    def forward(ctx, trans_acts, pred_acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda=0.0, validate=True):
        _certify(trans_acts, pred_acts, labels, act_lens, label_lens, validate)
        lib = _lib.lib()
        B, T, V = trans_acts.shape
        U = pred_acts.shape[1]
        need_grad = trans_acts.requires_grad or pred_acts.requires_grad
        dev = trans_acts.device
        with torch.cuda.device(dev):
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            ws = torch.empty(_lib.workspace_bytes_add(T, U, B), dtype=torch.uint8, device=dev)
            opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0,
                                   stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=int(blank),
                                   maxT=T, maxU=U, batch_first=True)
            lab_ptr = labels.data_ptr() if labels.numel() else costs.data_ptr()   # maxU == 1: never read
            st = lib.compute_rnnt_loss_add_fwd_dt(trans_acts.data_ptr(), pred_acts.data_ptr(), lab_ptr,
                                                  label_lens.data_ptr(), act_lens.data_ptr(), V, B,
                                                  costs.data_ptr(), ws.data_ptr(), opt, _DT[trans_acts.dtype],
                                                  1 if need_grad else 0, float(fastemit_lambda))
            _lib.check(st, "compute_rnnt_loss_add_fwd")
            ws.record_stream(torch.cuda.current_stream(dev))
        ctx.save_for_backward(trans_acts, pred_acts, labels, act_lens, label_lens)
        ctx.workspace = ws if need_grad else None
        ctx.blank = int(blank)
        ctx.mean_scale = 1.0 / B if reduction == "mean" else 1.0
        if reduction in ("sum", "mean"):
            costs = costs.sum().unsqueeze_(-1)
            if reduction == "mean":
                costs /= B
        return costs

    @staticmethod
    def backward(ctx, grad_output):
        trans_acts, pred_acts, labels, act_lens, label_lens = ctx.saved_tensors
        lib = _lib.lib()
        B, T, V = trans_acts.shape
        U = pred_acts.shape[1]
        dev = trans_acts.device
        with torch.cuda.device(dev):
            scale = (grad_output.reshape(-1).to(device=dev, dtype=torch.float32) * ctx.mean_scale).expand(B).contiguous()
            df = torch.empty_like(trans_acts)
            dg = torch.empty_like(pred_acts)
            opt = _lib.rnntOptions(loc=_lib.RNNT_GPU, num_threads=0,
                                   stream=torch.cuda.current_stream(dev).cuda_stream, blank_label=ctx.blank,
                                   maxT=T, maxU=U, batch_first=True)
            lab_ptr = labels.data_ptr() if labels.numel() else scale.data_ptr()
            st = lib.compute_rnnt_loss_add_bwd_dt(trans_acts.data_ptr(), pred_acts.data_ptr(), df.data_ptr(),
                                                  dg.data_ptr(), scale.data_ptr(), lab_ptr, label_lens.data_ptr(),
                                                  act_lens.data_ptr(), V, B, ctx.workspace.data_ptr(), opt,
                                                  _DT[trans_acts.dtype])
            _lib.check(st, "compute_rnnt_loss_add_bwd")
            ctx.workspace.record_stream(torch.cuda.current_stream(dev))
        return df, dg, None, None, None, None, None, None, None


_REDUCTIONS = {"none": 0, "sum": 1, "mean": 2}


def rnnt_loss_add(trans_acts, pred_acts, labels, act_lens, label_lens, blank=0, reduction="mean",
                  fastemit_lambda=0.0, validate=True):
    """RNN-T loss of the additive joint ``trans_acts[:, :, None] + pred_acts[:, None]`` without
    forming it.  Arguments as `rnnt_loss`, with the two activations instead of the joint tensor
    (float32, bfloat16 or float16; the loss is float32, the gradients have the activations' dtype).

    With the compiled extension module loaded (`warp_rnnt.binding() == "ext"`) the whole loss is its C++
    autograd function `rnnt_loss_add` (csrc/binding.cpp: the same checks, allocations and the two
    library calls without returning to Python); `_RNNTAdd` below is the ctypes twin.

    validate=False (as in `rnnt_loss`) skips the two checks that need the VALUES of the lengths (T == max(act_lens),
    U == max(label_lens) + 1: a device-to-host read and a synchronisation per call); without them the call only
    enqueues -- forward AND backward can then be captured in a HIP graph with the rest of a training step
    (tests/test_gpu_graph_step.py)."""
    from . import warp_rnnt
    ext = getattr(warp_rnnt, "_EXT", None)
    if ext is not None and reduction in _REDUCTIONS and isinstance(trans_acts, torch.Tensor) and trans_acts.is_cuda:
        return ext.rnnt_loss_add(trans_acts, pred_acts, labels, act_lens, label_lens, int(blank), _REDUCTIONS[reduction],
                                 float(fastemit_lambda), bool(validate))
    return _RNNTAdd.apply(trans_acts, pred_acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, validate)


class RNNTLossAdd(Module):
    def __init__(self, blank=0, reduction="mean", fastemit_lambda=0.0, validate=True):
        super().__init__()
        self.blank, self.reduction, self.fastemit_lambda, self.validate = blank, reduction, fastemit_lambda, validate

    def forward(self, trans_acts, pred_acts, labels, act_lens, label_lens):
        return rnnt_loss_add(trans_acts, pred_acts, labels, act_lens, label_lens, self.blank, self.reduction,
                             self.fastemit_lambda, self.validate)
