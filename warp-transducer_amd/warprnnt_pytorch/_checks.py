"""Argument validation shared by the loss wrappers (`RNNTLoss`, `RNNTLossAdd`, `RNNTLossPacked`).

Behavioural contract = the reference binding's (pytorch_binding/warprnnt_pytorch/__init__.py:103-140): the same
checks in the same ORDER (so the first violated one decides the exception), the same exception types and the
same message texts -- including its spelling "lenghts" in two of them, which callers may be matching on.
"""
import torch


def check_type(var, t, name):
    if var.dtype is not t:
        raise TypeError("%s must be %s" % (name, t))


def check_contiguous(var, name):
    if not var.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def check_dim(var, dim, name):
    if var.dim() != dim:
        raise ValueError("%s must be %dD" % (name, dim))


def certify_inputs(log_probs, labels, lengths, label_lengths, read_lengths=True):
    """(N,T,U,V) activations, (N,U-1) int32 labels, (N,) int32 lengths: dtype, contiguity, one length per sample,
    ranks, and T == max(lengths), U == max(label_lengths) + 1 (one device-to-host read for both).
    read_lengths=False skips that last pair of checks -- the only ones that need the VALUES of the lengths, i.e. a
    device-to-host read and a stream synchronisation per call when they live on the GPU (`RNNTLoss(validate=False)`)."""
    arg = {"log_probs": log_probs, "labels": labels, "lengths": lengths, "label_lengths": label_lengths}
    for name in ("labels", "label_lengths", "lengths"):
        check_type(arg[name], torch.int32, name)
    for name in ("log_probs", "labels", "label_lengths", "lengths"):
        check_contiguous(arg[name], name)
    batch = log_probs.shape[0]
    if lengths.shape[0] != batch:
        raise ValueError("must have a length per example.")
    if label_lengths.shape[0] != batch:
        raise ValueError("must have a label length per example.")
    for name, rank, shown in (("log_probs", 4, "log_probs"), ("labels", 2, "labels"),
                              ("lengths", 1, "lenghts"), ("label_lengths", 1, "label_lenghts")):
        check_dim(arg[name], rank, shown)
    if not read_lengths:
        return
    # the reference reads the two maxima back one after the other (two device synchronisations per forward when
    # the lengths live on the GPU); one reduction over both vectors and one read keep its checks and their order
    if lengths.device == label_lengths.device:
        max_t, max_l = torch.stack((lengths, label_lengths)).amax(1).tolist()
    else:
        max_t, max_l = int(lengths.max()), int(label_lengths.max())
    if log_probs.shape[1] != max_t:
        raise ValueError("Input length mismatch")
    if log_probs.shape[2] != max_l + 1:
        raise ValueError("Output length mismatch")


def check_gpu_arguments(acts, labels=None, lengths=None, label_lengths=None, workspace=None, workspace_bytes=None):
    """The GPU location dereferences labels and both length vectors on the device of the activations: a host tensor (or one
    of another GPU) there would be a GPU fault, not an exception.  A caller-owned workspace must live on that device too
    and hold at least what get_workspace_size asks for TODAY (the layout is private and has grown between versions)."""
    for name, t in (("labels", labels), ("lengths", lengths), ("label_lengths", label_lengths)):
        if t is None or (name == "labels" and t.numel() == 0):
            continue
        if not t.is_cuda or t.device != acts.device:
            raise ValueError("%s must be on the device of the activations (%s), got %s" % (name, acts.device, t.device))
    if workspace is not None:
        if not workspace.is_cuda or workspace.device != acts.device:
            raise ValueError("workspace must be on the device of the activations")
        have = workspace.numel() * workspace.element_size()
        if not workspace.is_contiguous() or (workspace_bytes is not None and have < workspace_bytes):
            raise ValueError("workspace of %d bytes, get_workspace_size asks for %s" % (have, workspace_bytes))
