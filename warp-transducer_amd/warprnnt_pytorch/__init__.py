"""warprnnt_pytorch -- drop-in for the reference's PyTorch binding, backed by the MI355X
library (libwarprnnt.so, hand-written HIP for gfx950).

Public surface identical to pytorch_binding/warprnnt_pytorch/__init__.py:
``rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction='mean')`` and
``RNNTLoss(blank=0, reduction='mean')`` (reference __init__.py:8,53-100), same input checks
and error types (``certify_inputs``, :103-140), same reduction semantics (:36-40), gradients
computed in forward and scaled in backward (:43-50).
"""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import warp_rnnt
from ._checks import certify_inputs, check_contiguous, check_dim, check_type

__all__ = ['rnnt_loss', 'RNNTLoss']

# GPU tensors go through compute_rnnt_loss_async (device costs, no sync) unless
# WARPRNNT_SYNC_API=1 asks for the reference's host-costs entry point compute_rnnt_loss.
import os as _os
_ASYNC_GPU = _os.environ.get("WARPRNNT_SYNC_API", "0") != "1"


class _RNNT(Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda=0.0, validate=True):
        """
        acts:       (batch, T, U, vocab) joint-network output; raw logits on the GPU,
                    log-probabilities on the CPU (the wrappers below apply log_softmax there)
        labels:     (batch, U-1) int32 targets, zero padded
        act_lens:   (batch,) int32 number of valid time steps per sample
        label_lens: (batch,) int32 number of valid labels per sample
        """
        is_cuda = acts.is_cuda
        certify_inputs(acts, labels, act_lens, label_lens, read_lengths=validate)

        minibatch_size = acts.size(0)
        cost_dtype = acts.dtype if acts.dtype in (torch.float32, torch.float64) else torch.float32
        ctx.two_phase = bool(is_cuda and _ASYNC_GPU)
        if fastemit_lambda and not ctx.two_phase:
            raise NotImplementedError("fastemit_lambda is an extension of the GPU two-phase route")
        if ctx.two_phase:
            # Two-phase route (compute_rnnt_loss_fwd / _bwd).  The reference computes the full gradient
            # tensor here, keeps it in ctx, divides it by N for 'mean' and multiplies it by grad_output
            # in backward (__init__.py:24,36-50): two extra read+write passes over the (B,T,U,V) tensor
            # and a tensor of that size alive between forward and backward.  Here forward leaves only
            # the workspace (lattice + coefficient table) behind; backward runs the gradient kernel
            # once with the 1/N and grad_output factors folded in.  Same values, costs stay on device.
            costs = torch.empty(minibatch_size, dtype=cost_dtype, device=acts.device)
            # (the workspace is allocated on the stream that is current here; gpu_rnnt_bwd records a different backward
            # stream with the allocator if there ever is one)
            ws = warp_rnnt.gpu_rnnt_fwd(acts, labels, act_lens, label_lens, costs, blank, acts.requires_grad,
                                        fastemit_lambda)
            ctx.save_for_backward(acts)
            ctx.workspace = ws if acts.requires_grad else None
            ctx.blank = blank
            ctx.mean_scale = 1.0 / minibatch_size if reduction == 'mean' else 1.0
            ctx.grads = None
            # one reduction kernel ('mean' = sum / N, as the reference divides: __init__.py:36-40)
            if reduction == 'sum':
                return costs.sum(0, keepdim=True)
            if reduction == 'mean':
                return costs.mean(0, keepdim=True)
            return costs
        else:
            # The library overwrites every element of grads (zeros in the padded region), so no
            # zero-fill is needed (the reference allocates zeros_like: __init__.py:24).
            grads = torch.empty_like(acts) if acts.requires_grad else torch.zeros(0).to(acts)
            loss_func = warp_rnnt.gpu_rnnt if is_cuda else warp_rnnt.cpu_rnnt
            # host, as the C-ABI requires; pinned for the GPU location (the library then writes it from the
            # lattice kernel instead of staging a pageable copy)
            costs = torch.zeros(minibatch_size, dtype=cost_dtype, pin_memory=bool(is_cuda))
            # cpu_rnnt / gpu_rnnt keep the reference extension module's behaviour for an unsupported dtype
            # (a line on stderr and -1, binding.cpp:46-81,111-153), which the reference wrapper ignores and
            # then returns a zero loss with zero gradients; here nothing would have been written at all, so
            # it is an error.
            if loss_func(acts, labels, act_lens, label_lens, costs, grads, blank, 0) != 0:
                raise TypeError("rnnt_loss: unsupported dtype %s for the %s location"
                                % (acts.dtype, "GPU" if is_cuda else "CPU"))

        if reduction in ['sum', 'mean']:
            costs = costs.sum().unsqueeze_(-1)
            if reduction == 'mean':
                costs /= minibatch_size
                if grads is not None:
                    grads /= minibatch_size

        costs = costs.to(acts.device)
        ctx.grads = grads

        return costs

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.two_phase:
            (acts,) = ctx.saved_tensors
            n = acts.size(0)
            sdt = torch.float64 if acts.dtype == torch.float64 else torch.float32
            g = grad_output.reshape(-1)
            if g.dtype != sdt or g.device != acts.device:
                g = g.to(device=acts.device, dtype=sdt)
            # the per-sample factor of the gradient kernel in ONE elementwise kernel (broadcast of a (1,) grad_output
            # for 'sum' / 'mean'; the 1/N of 'mean' folded in)
            scale = g.expand(n) * ctx.mean_scale
            grads = torch.empty_like(acts)
            warp_rnnt.gpu_rnnt_bwd(acts, grads, scale, ctx.workspace, ctx.blank)
            return grads, None, None, None, None, None, None, None
        # out of place (the reference scales the saved tensor in place, __init__.py:47-50, so a second
        # backward through a retained graph compounds the factors and gradcheck fails).  Cost: one more
        # (B,T,U,V) tensor alive during backward on THIS route -- the CPU location and WARPRNNT_SYNC_API=1;
        # the default GPU route above keeps no gradient tensor at all between forward and backward.
        grad_output = grad_output.view(-1, 1, 1, 1).to(ctx.grads)
        return ctx.grads * grad_output, None, None, None, None, None, None, None


_REDUCTIONS = {'none': 0, 'sum': 1, 'mean': 2}


def _apply(acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, validate):
    """GPU tensors on the default (two-phase) route go through the compiled module's C++ autograd function when it is
    there (warp_rnnt.binding() == 'ext': checks, allocations, both library calls and the reduction without returning to
    Python); everything else -- the CPU location, WARPRNNT_SYNC_API=1, the ctypes binding -- through `_RNNT` above.
    Same values either way (tests run both)."""
    ext = getattr(warp_rnnt, "_EXT", None)        # (tests swap in the reference's own extension module, which has none)
    if ext is not None and acts.is_cuda and _ASYNC_GPU and reduction in _REDUCTIONS:
        return ext.rnnt_loss(acts, labels, act_lens, label_lens, int(blank), _REDUCTIONS[reduction], float(fastemit_lambda),
                             bool(validate))
    return _RNNT.apply(acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, validate)


def rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction='mean', fastemit_lambda=0.0, validate=True):
    """RNN Transducer loss.

    Args:
        acts: (batch, T, U, vocab) tensor, output of the joint network
        labels: (batch, U-1) int32 tensor of targets, zero padded
        act_lens: (batch,) int32 tensor, valid time steps of each sample
        label_lens: (batch,) int32 tensor, valid labels of each sample
        blank (int, optional): blank label. Default: 0.
        reduction (string, optional): 'none' | 'mean' | 'sum'. 'none': per-sample losses;
            'sum': summed over the batch (shape (1,)); 'mean': the sum divided by the batch
            size. Default: 'mean'
        fastemit_lambda (float, optional): extension, GPU only -- FastEmit regularisation (Yu et al.
            2021): the gradient of every label transition is scaled by (1 + fastemit_lambda); the
            returned loss is the plain negative log-likelihood. Default: 0.0 (the reference's loss)
        validate (bool, optional): extension -- False skips the two checks that need the VALUES of the lengths
            (T == max(act_lens), U == max(label_lens) + 1: reference __init__.py:134-139), which cost a device-to-host
            read and a stream synchronisation per call when the lengths live on the GPU; everything else is still
            checked, and lengths that do not fit the tensor make the sample's loss NaN. Default: True
    """
    if not acts.is_cuda:
        acts = torch.nn.functional.log_softmax(acts, -1)
    return _apply(acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, validate)


class RNNTLoss(Module):
    """
    Parameters:
        blank (int, optional): blank label. Default: 0.
        reduction (string, optional): 'none' | 'mean' | 'sum' (see `rnnt_loss`). Default: 'mean'
        fastemit_lambda (float, optional): extension, GPU only (see `rnnt_loss`). Default: 0.0
        validate (bool, optional): extension (see `rnnt_loss`): False = no device-to-host read of the lengths per
            call. Default: True (the reference's behaviour)
    """

    def __init__(self, blank=0, reduction='mean', fastemit_lambda=0.0, validate=True):
        super(RNNTLoss, self).__init__()
        self.blank = blank
        self.reduction = reduction
        self.fastemit_lambda = fastemit_lambda
        self.validate = validate
        self.loss = _apply

    def forward(self, acts, labels, act_lens, label_lens):
        if not acts.is_cuda:
            # The CPU location of the library takes log-probabilities; log_softmax runs inside
            # the kernels only on the GPU (reference __init__.py:95-98).
            acts = torch.nn.functional.log_softmax(acts, -1)
        return self.loss(acts, labels, act_lens, label_lens, self.blank, self.reduction, self.fastemit_lambda, self.validate)
