"""warprnnt_tensorflow -- the reference's TensorFlow API (tensorflow_binding/warprnnt_tensorflow/__init__.py:9-47)
over this library: ``rnnt_loss(acts, labels, input_lengths, label_lengths, blank_label=0)`` with the gradient
registered for the "WarpRNNT" op.  Needs tensorflow-rocm and kernels.so built by ../build.sh.  Device placement decides the
contract, as in the reference (tensorflow_binding/tests/test_warprnnt_op.py:20): on the GPU `acts` are raw logits and the gradient
is the dense d/d(logits); the CPU kernel takes LOG-PROBABILITIES (apply tf.nn.log_softmax first, the chain rule then gives the same
dense gradient) -- the GPU location of the library itself never falls back to the host."""
import os

import tensorflow as tf

_kernels = tf.load_op_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels.so"))

__all__ = ["rnnt_loss"]


def _runs_on_cpu(acts):
    """Will the op for `acts` be placed on the CPU kernel?  No GPU visible, or an explicit CPU device scope / tensor."""
    if not tf.config.list_physical_devices("GPU"):
        return True
    dev = getattr(acts, "device", "") or ""
    return "CPU" in dev.upper() and "GPU" not in dev.upper()


def rnnt_loss(acts, labels, input_lengths, label_lengths, blank_label=0):
    """RNN-T loss of joint-network LOGITS.

    acts: (B, T, U, V) float32 logits; labels: (B, U-1) int32, zero padded; input_lengths, label_lengths: (B,) int32.
    Returns the (B,) negative log-likelihoods.  On the GPU the log-softmax is applied inside the kernels.  The CPU kernel of
    the op takes LOG-PROBABILITIES (the reference's CPU contract): when the op will be placed there -- no GPU visible, or a CPU
    device scope -- tf.nn.log_softmax is applied here first, so the same call returns the same loss and, through the chain
    rule, the same dense logit gradient on either device instead of a silently wrong one.  Callers of the raw
    op (`_kernels.warp_rnnt`) on the CPU apply it themselves, as the reference's test does
    (tensorflow_binding/tests/test_warprnnt_op.py:20)."""
    if _runs_on_cpu(acts):
        acts = tf.nn.log_softmax(acts, axis=-1)
    costs, _ = _kernels.warp_rnnt(acts, labels, input_lengths, label_lengths, blank_label=blank_label)
    return costs


@tf.RegisterGradient("WarpRNNT")
def _rnnt_loss_grad(op, grad_costs, _unused_grad_of_grads):
    # the op's second output IS d(cost_b)/d(acts_b); chain with the incoming per-sample gradient
    scale = tf.reshape(grad_costs, (-1, 1, 1, 1))
    return [scale * op.outputs[1], None, None, None]
