"""Batch-sharded RNN-T loss over the GPUs of one node (one process per GPU, RCCL over xGMI).

Not in the reference (it has no multi-device layer: SURVEY.md 2.1/8e).  Samples are independent,
so every rank runs the single-GPU hot path on its own contiguous slab of the batch with its own
workspace and stream; gradients stay local (data parallel).  The data path needs exactly ONE
collective: an all-reduce(sum) of the 2-element vector [local summed loss, local sample count]
('sum'/'mean': shards may be ragged, 'mean' divides by the GLOBAL sample count), or an all-gather of the
per-sample costs ('none'; ragged shards are padded to the largest, their sizes gathered first).  The payload is 8 bytes, so
the collective is latency-bound and is enqueued on the compute stream -- nothing is staged
through the host.  On CPU tensors (tests: gloo, world_size 2) the same code runs the library's
RNNT_CPU location.
"""
import torch
import torch.distributed as dist
from torch.autograd import Function
from torch.nn import Module

from . import certify_inputs, warp_rnnt

__all__ = ["sharded_rnnt_loss", "ShardedRNNTLoss"]


class _ShardedRNNT(Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction, group):
        certify_inputs(acts, labels, act_lens, label_lens)
        n = acts.size(0)
        cost_dtype = acts.dtype if acts.dtype in (torch.float32, torch.float64) else torch.float32
        ctx.on_gpu = acts.is_cuda
        if acts.is_cuda:
            # forward phase only: costs stay on the device, the workspace carries the coefficient
            # table to backward (compute_rnnt_loss_fwd / _bwd), no gradient tensor is kept
            costs = torch.empty(n, dtype=cost_dtype, device=acts.device)
            ws = warp_rnnt.gpu_rnnt_fwd(acts, labels, act_lens, label_lens, costs, blank, acts.requires_grad)
            ctx.save_for_backward(acts)         # (the workspace's stream bookkeeping: warp_rnnt.gpu_rnnt_fwd / _bwd)
            ctx.workspace, ctx.blank, grads = ws, blank, None
        else:
            costs = torch.zeros(n, dtype=cost_dtype)
            grads = torch.empty_like(acts) if acts.requires_grad else torch.zeros(0).to(acts)
            if warp_rnnt.cpu_rnnt(acts, labels, act_lens, label_lens, costs, grads, blank, 0) != 0:
                raise TypeError("sharded_rnnt_loss: unsupported dtype %s for the CPU location" % acts.dtype)
        distributed = dist.is_available() and dist.is_initialized()
        ctx.scale = 1.0
        if reduction == "none":
            ctx.rank_offset, ctx.local_n = 0, n
            out = costs
            if distributed:
                world, rank = dist.get_world_size(group), dist.get_rank(group)
                # shards may be ragged (a last shard with fewer samples): one tiny all-gather of the shard sizes first,
                # then the per-sample costs padded to the largest shard (all_gather wants equal shapes)
                mine = torch.full((1,), n, dtype=torch.int64, device=costs.device)
                sizes = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(sizes, mine, group=group)
                sizes = [int(t.item()) for t in sizes]
                top = max(sizes)
                padded = costs if n == top else torch.cat([costs, costs.new_zeros(top - n)])
                parts = [torch.empty_like(padded) for _ in range(world)]
                dist.all_gather(parts, padded, group=group)
                out = torch.cat([p[:k] for p, k in zip(parts, sizes)])
                ctx.rank_offset = sum(sizes[:rank])
        else:
            # [local summed loss, local sample count] built on the device: one fill and one reduction, no host-to-device
            # copy and no synchronisation in front of the collective
            packed = torch.full((2,), float(n), dtype=torch.float64, device=costs.device)
            torch.sum(costs, dim=0, keepdim=True, dtype=torch.float64, out=packed[0:1])
            if distributed:
                dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)   # the single collective
            out = packed[0:1].to(costs.dtype)
            if reduction == "mean":
                out = out / packed[1].to(costs.dtype)
                ctx.scale = 1.0 / packed[1]          # 1 / GLOBAL batch size (a 0-dim device tensor)
        ctx.grads = grads
        ctx.reduction = reduction
        return out

    @staticmethod
    def backward(ctx, grad_output):
        g = grad_output
        if ctx.reduction == "none":
            g = g[ctx.rank_offset:ctx.rank_offset + ctx.local_n]
        if ctx.on_gpu:
            (acts,) = ctx.saved_tensors
            sdt = torch.float64 if acts.dtype == torch.float64 else torch.float32
            scale = (g.reshape(-1).to(device=acts.device, dtype=sdt) * ctx.scale).to(sdt).expand(acts.size(0)).contiguous()
            grads = torch.empty_like(acts)
            warp_rnnt.gpu_rnnt_bwd(acts, grads, scale, ctx.workspace, ctx.blank)
            return grads, None, None, None, None, None, None
        g = g.reshape(-1, 1, 1, 1).to(ctx.grads)
        if not isinstance(ctx.scale, float) or ctx.scale != 1.0:
            g = g * ctx.scale
        return ctx.grads * g.to(ctx.grads.dtype), None, None, None, None, None, None


def sharded_rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction="mean", group=None):
    """`rnnt_loss` over a batch sharded across the ranks of `group`; every rank passes its own
    shard and receives the loss of the GLOBAL batch ('mean' divides by the global batch size)."""
    if not acts.is_cuda:
        acts = torch.nn.functional.log_softmax(acts, -1)
    return _ShardedRNNT.apply(acts, labels, act_lens, label_lens, blank, reduction, group)


class ShardedRNNTLoss(Module):
    def __init__(self, blank=0, reduction="mean", group=None):
        super().__init__()
        self.blank, self.reduction, self.group = blank, reduction, group

    def forward(self, acts, labels, act_lens, label_lens):
        return sharded_rnnt_loss(acts, labels, act_lens, label_lens, self.blank, self.reduction, self.group)
