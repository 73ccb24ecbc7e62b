"""Batch-sharded RNN-T loss over the GPUs of one node (one process per GPU, RCCL over xGMI).

Not in the reference (it has no multi-device layer: SURVEY.md 2.1/8e).  Samples are independent,
so every rank runs the single-GPU hot path on its own contiguous slab of the batch with its own
workspace and stream; gradients stay local (data parallel).  The data path needs exactly ONE
collective: an all-reduce(sum) of the 2-element vector [local summed loss, local sample count]
('sum'/'mean': shards may be ragged, 'mean' divides by the GLOBAL sample count), or ONE all-gather of
[shard size, per-sample costs padded to the shard capacity] ('none'; no size exchange, no host read).  The payload is tiny, so
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
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction, group, global_batch):
        distributed = dist.is_available() and dist.is_initialized()
        n = acts.size(0)
        cost_dtype = acts.dtype if acts.dtype in (torch.float32, torch.float64) else torch.float32
        ctx.on_gpu = acts.is_cuda
        # ALL RANKS OR NONE (as compute_rnnt_loss_sharded, include/rnnt.h): a rank whose LOCAL part fails -- bad arguments
        # for its shard, an unsupported dtype -- still joins the one collective, with NaNs, so that its peers are not left
        # blocked and every rank's loss is NaN; it then raises its own error.
        failure, grads, ws = None, None, None
        try:
            certify_inputs(acts, labels, act_lens, label_lens)
            if acts.is_cuda:
                # forward phase only: costs stay on the device, the workspace carries the coefficient
                # table to backward (compute_rnnt_loss_fwd / _bwd), no gradient tensor is kept
                costs = torch.empty(n, dtype=cost_dtype, device=acts.device)
                ws = warp_rnnt.gpu_rnnt_fwd(acts, labels, act_lens, label_lens, costs, blank, acts.requires_grad)
            else:
                costs = torch.zeros(n, dtype=cost_dtype)
                grads = torch.empty_like(acts) if acts.requires_grad else torch.zeros(0).to(acts)
                if warp_rnnt.cpu_rnnt(acts, labels, act_lens, label_lens, costs, grads, blank, 0) != 0:
                    raise TypeError("sharded_rnnt_loss: unsupported dtype %s for the CPU location" % acts.dtype)
        except Exception as exc:                                    # noqa: BLE001 -- re-raised below, after the collective
            if not distributed:
                raise
            failure = exc
            costs = torch.full((n,), float("nan"), dtype=cost_dtype, device=acts.device)
        if acts.is_cuda and failure is None:
            ctx.save_for_backward(acts)         # (the workspace's stream bookkeeping: warp_rnnt.gpu_rnnt_fwd / _bwd)
            ctx.workspace, ctx.blank = ws, blank
        ctx.scale = 1.0
        if reduction == "none":
            ctx.rank_offset, ctx.local_n = 0, n
            out = costs
            if distributed:
                # ONE collective and no host synchronisation: every rank contributes [n, its costs padded to the
                # shard capacity]; the capacity is known WITHOUT communication -- ceil(global_batch / world) when the caller
                # states the global batch (ragged shards allowed), else this rank's own n (equal shards assumed, and checked
                # on the device: a mismatch turns every cost into NaN instead of a silently mis-sliced tensor).  The global
                # cost vector is assembled by a device-side scatter; nothing reads a size back to the host.
                world, rank = dist.get_world_size(group), dist.get_rank(group)
                cap = n if global_batch is None else -(-int(global_batch) // world)
                total = n * world if global_batch is None else int(global_batch)
                payload = costs.new_zeros(cap + 1)
                payload[0] = n
                if n > cap:
                    failure = failure or ValueError("sharded_rnnt_loss: this rank's shard has %d samples, more than "
                                                    "ceil(global_batch / world) = %d" % (n, cap))
                    payload[1:] = float("nan")
                else:
                    payload[1:1 + n] = costs
                gathered = costs.new_empty((world, cap + 1))
                dist.all_gather_into_tensor(gathered.view(-1), payload, group=group)      # the single collective
                sizes = gathered[:, 0]
                offsets = torch.cumsum(sizes, 0) - sizes
                slot = torch.arange(cap, device=costs.device, dtype=sizes.dtype)
                dest = torch.where(slot[None, :] < sizes[:, None], offsets[:, None] + slot[None, :],
                                   torch.full((), float(total), device=costs.device, dtype=sizes.dtype))
                dest = dest.clamp(0, total).long()
                ext = costs.new_full((total + 1,), float("nan"))    # a slot nobody fills stays NaN: sizes that do not add up show
                ext.scatter_(0, dest.reshape(-1), gathered[:, 1:].reshape(-1))
                out = ext[:total]
                consistent = sizes.sum() == total
                if global_batch is None:
                    consistent = consistent & (sizes == n).all()
                out = torch.where(consistent, out, torch.full_like(out, float("nan")))
                ctx.rank_offset = offsets[rank].long()               # a 0-dim DEVICE tensor (backward gathers with it)
        else:
            # [local summed loss, local sample count] built on the device: one fill and one reduction, no host-to-device
            # copy and no synchronisation in front of the collective
            packed = torch.full((2,), float(n), dtype=torch.float64, device=costs.device)
            torch.sum(costs, dim=0, keepdim=True, dtype=torch.float64, out=packed[0:1])
            if failure is not None:
                packed.fill_(float("nan"))
            if distributed:
                dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)   # the single collective
            out = packed[0:1].to(costs.dtype)
            if reduction == "mean":
                out = out / packed[1].to(costs.dtype)
                ctx.scale = 1.0 / packed[1]          # 1 / GLOBAL batch size (a 0-dim device tensor)
        if failure is not None:
            raise failure
        ctx.grads = grads
        ctx.reduction = reduction
        return out

    @staticmethod
    def backward(ctx, grad_output):
        g = grad_output
        if ctx.reduction == "none":
            if isinstance(ctx.rank_offset, int):
                g = g[ctx.rank_offset:ctx.rank_offset + ctx.local_n]
            else:                                                    # device offset: a gather, no host read
                g = g.index_select(0, torch.arange(ctx.local_n, device=g.device) + ctx.rank_offset.to(g.device))
        if ctx.on_gpu:
            (acts,) = ctx.saved_tensors
            sdt = torch.float64 if acts.dtype == torch.float64 else torch.float32
            scale = (g.reshape(-1).to(device=acts.device, dtype=sdt) * ctx.scale).to(sdt).expand(acts.size(0)).contiguous()
            grads = torch.empty_like(acts)
            warp_rnnt.gpu_rnnt_bwd(acts, grads, scale, ctx.workspace, ctx.blank)
            return grads, None, None, None, None, None, None, None
        g = g.reshape(-1, 1, 1, 1).to(ctx.grads)
        if not isinstance(ctx.scale, float) or ctx.scale != 1.0:
            g = g * ctx.scale
        return ctx.grads * g.to(ctx.grads.dtype), None, None, None, None, None, None, None


def sharded_rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction="mean", group=None, global_batch=None):
    """`rnnt_loss` over a batch sharded across the ranks of `group`; every rank passes its own
    shard and receives the loss of the GLOBAL batch ('mean' divides by the global batch size).
    reduction='none' returns the per-sample costs of the global batch in rank order through ONE all-gather and no host
    synchronisation: with `global_batch` (the job's sample count, the same on every rank) shards may be ragged, none larger
    than ceil(global_batch / world); without it every rank must pass the same number of samples (a mismatch makes every cost
    NaN).  'sum' / 'mean' take ragged shards as they are (the all-reduced pair carries the count)."""
    if not acts.is_cuda:
        acts = torch.nn.functional.log_softmax(acts, -1)
    return _ShardedRNNT.apply(acts, labels, act_lens, label_lens, blank, reduction, group, global_batch)


class ShardedRNNTLoss(Module):
    def __init__(self, blank=0, reduction="mean", group=None, global_batch=None):
        super().__init__()
        self.blank, self.reduction, self.group, self.global_batch = blank, reduction, group, global_batch

    def forward(self, acts, labels, act_lens, label_lens):
        return sharded_rnnt_loss(acts, labels, act_lens, label_lens, self.blank, self.reduction, self.group, self.global_batch)
