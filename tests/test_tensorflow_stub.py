"""The TensorFlow op source (warp-transducer_amd/tensorflow_binding/warprnnt_op.cc) meets a compiler and its kernels RUN --
against a stand-in for the TensorFlow declarations it uses (tests/tf_stub/: test infrastructure, not TensorFlow), because
this image has no TensorFlow and no index to install one from.

What this pins: the file is well-formed C++ against the TF 2.x kernel API surface it touches (OpKernel / OpKernelContext /
Tensor / TensorShape / REGISTER_OP / REGISTER_KERNEL_BUILDER / shape inference); the op is registered under the reference's
name with its inputs, attr and outputs (tensorflow_binding/src/warprnnt_op.cc:13-20) and a shape function that rejects a
wrong rank; a CPU kernel and a GPU kernel are registered (reference :142-161, :165-187); and each kernel's Compute() --
shape checks, output / workspace allocation, options, the C-ABI call -- produces the reference's golden numbers
(tests/test_cpu.cpp:79-109 sparse log-prob gradients for the CPU kernel; tensorflow_binding/tests/test_warprnnt_op.py:68-79 =
pytorch_binding/test/test.py:61-78 dense logit gradients for the GPU kernel).  What it cannot pin: the real headers' ABI,
`tf.load_op_library`, the registered Python gradient -- tests/test_tensorflow_binding.py does that where TensorFlow exists.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.golden import literals as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "tf_stub")
LIBDIR = os.path.join(ROOT, "warp-transducer_amd", "lib")


def build(tmp_path, device_memory):
    exe = str(tmp_path / ("run_op_gpu" if device_memory else "run_op_cpu"))
    if device_memory:
        cmd = ["hipcc", "--offload-arch=gfx950", "-DTF_STUB_DEVICE_MEMORY=1"]
    else:
        cmd = ["g++"]
    cmd += ["-std=c++17", "-O1", "-Wall", "-I" + STUB, "-I" + os.path.join(ROOT, "include"), os.path.join(STUB, "run_op.cpp"), "-o", exe,
            "-L" + LIBDIR, "-lwarprnnt", "-Wl,-rpath," + LIBDIR]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    assert "warning" not in out.stderr, out.stderr[-3000:]          # the op source compiles clean under -Wall against the stand-in
    return exe


def run(exe, device, blank, acts, labels, tl, ll):
    B, T, U, V = acts.shape
    text = "%s %d %d %d %d %d\n" % (device, blank, B, T, U, V)
    text += " ".join("%.9g" % v for v in acts.astype(np.float32).ravel()) + "\n"
    text += " ".join(str(int(v)) for v in np.asarray(labels).ravel()) + "\n"
    text += " ".join(str(int(v)) for v in tl) + "\n" + " ".join(str(int(v)) for v in ll) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.splitlines()
    head = {l.split()[0]: l for l in lines if l and l.split()[0] in ("op", "shapes", "shapes_bad_rank", "status")}
    values = np.array([float(l) for l in lines if l and l.split()[0] not in head], dtype=np.float64)
    return head, values[:B], values[B:].reshape(acts.shape) if values.size > B else None


def log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def test_cpu_kernel_runs_the_reference_golden_vectors(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libwarprnnt.so")):
        pytest.skip("libwarprnnt.so not built")
    exe = build(tmp_path, device_memory=False)
    head, costs, grads = run(exe, "CPU", 0, log_softmax(G.OPTIONS_ACTS_6DP.astype(np.float64)), G.OPTIONS_LABELS, [4, 4], [2, 2])
    assert head["op"].split() == ["op", "WarpRNNT", "inputs", "4", "outputs", "2", "attrs", "1"]
    assert head["shapes"].split()[1] == "OK" and head["shapes"].split()[-3:] == ["1", "grads_rank", "4"]
    assert head["shapes_bad_rank"].split()[1] == "rejected"
    assert head["status"] == "status OK"
    assert np.abs(costs - G.OPTIONS_COSTS).max() < 1e-4                                   # tests/test_cpu.cpp:107-109
    assert np.abs(grads - G.OPTIONS_LOGPROB_GRADS.reshape(grads.shape)).max() < 1e-4      # tests/test_cpu.cpp:94-105 (sparse, wrt log-probs)
    # small_test (tests/test_cpu.cpp:12-71), forward value
    _, c1, _ = run(exe, "CPU", 0, log_softmax(G.SMALL_ACTS.astype(np.float64)), G.SMALL_LABELS, [2], [2])
    assert abs(c1[0] - G.SMALL_COST) < 1e-4
    # labels equal to the blank are legal input (the CPU location keeps the CPU reference's answer): status OK
    head, _, _ = run(exe, "CPU", 0, log_softmax(G.SMALL_ACTS.astype(np.float64)), np.zeros((1, 2), np.int32), [2], [2])
    assert head["status"] == "status OK"
    # the library's own argument checks answer through the context's status, not a crash
    bad_blank, _, _ = run(exe, "CPU", 7, log_softmax(G.SMALL_ACTS.astype(np.float64)), G.SMALL_LABELS, [2], [2])
    assert "invalid value" in bad_blank["status"]                                          # blank outside the vocabulary: the library's status, as text


@pytest.mark.gpu
def test_gpu_kernel_runs_the_reference_golden_vectors(tmp_path):
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc to build the harness with device tensors")
    exe = build(tmp_path, device_memory=True)
    # tensorflow_binding/tests/test_warprnnt_op.py:20-28: raw logits on the GPU, costs and DENSE logit gradients
    head, costs, grads = run(exe, "GPU", 0, G.SMALL_ACTS, G.SMALL_LABELS, [2], [2])
    assert head["status"] == "status OK"
    assert abs(costs[0] - G.SMALL_COST) < 1e-4 and np.abs(grads - G.SMALL_GRADS).max() < 1e-5
    head, costs, grads = run(exe, "GPU", 0, G.BIG_ACTS, G.OPTIONS_LABELS, [4, 4], [2, 2])   # _test_multiple_batches (:67-79)
    assert head["status"] == "status OK"
    assert np.abs(costs - G.OPTIONS_COSTS).max() < 1e-4 and np.allclose(grads, G.BIG_GRADS, rtol=1e-3, atol=1e-6)
