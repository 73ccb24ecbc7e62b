"""The TensorFlow binding (warp-transducer_amd/tensorflow_binding): runs only where tensorflow-rocm is installed
and kernels.so has been built (build.sh); this repository's image has no TensorFlow, so here it is skipped."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
tf = pytest.importorskip("tensorflow")
BINDING = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "warp-transducer_amd",
                       "tensorflow_binding")


def test_loss_and_gradient_against_the_oracle(oracle):
    if not os.path.exists(os.path.join(BINDING, "warprnnt_tensorflow", "kernels.so")):
        pytest.skip("kernels.so not built (tensorflow_binding/build.sh)")
    sys.path.insert(0, BINDING)
    from warprnnt_tensorflow import rnnt_loss
    rng = np.random.default_rng(0)
    B, T, U, V, blank = 3, 9, 5, 11, 2
    acts = rng.standard_normal((B, T, U, V)).astype(np.float32)
    labels = rng.integers(0, V, size=(B, U - 1)).astype(np.int32)
    labels[labels == blank] = (blank + 1) % V
    tl = np.array([9, 4, 7], dtype=np.int32)
    ll = np.array([4, 2, 0], dtype=np.int32)
    weights = np.array([1.0, -0.5, 2.0], dtype=np.float32)
    with tf.device("/GPU:0"):
        x = tf.Variable(acts)
        with tf.GradientTape() as tape:
            costs = rnnt_loss(x, tf.constant(labels), tf.constant(tl), tf.constant(ll), blank_label=blank)
            total = tf.reduce_sum(costs * weights)
        grads = tape.gradient(total, x)
    ref_c, ref_g = oracle.rnnt_logits(acts.astype(np.float64), labels, tl, ll, blank)
    assert np.allclose(costs.numpy(), ref_c, rtol=1e-4)
    assert np.allclose(grads.numpy(), ref_g * weights[:, None, None, None], rtol=1e-3, atol=1e-4)
