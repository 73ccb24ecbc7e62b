"""The REFERENCE's own test and binding sources, compiled UNMODIFIED against this repo's include/rnnt.h and
libwarprnnt.so -- the drop-in claim of INTEGRATION.md 1-2 as a test.

  test_reference_test_cpu_cpp     /root/reference/tests/test_cpu.cpp + random.cpp (small_test, options_test,
                                  inf_test, grad_check through the C-ABI, RNNT_CPU) -> "Tests pass"
                                  (tests/test_cpu.cpp:382-392).  One token is patched on the fly: `float
                                  numeric_grad` has no return statement (tests/test_cpu.cpp:242-285), which is
                                  undefined behaviour and crashes under GCC 11.
  test_reference_binding_and_test_py
                                  /root/reference/pytorch_binding/src/binding.cpp built by
                                  oracle/build_ref_binding.py, imported by the reference's OWN
                                  warprnnt_pytorch/__init__.py, driven by the reference's OWN
                                  pytorch_binding/test/test.py (small_test, big_test: test.py:52-161).
  test_reference_gpu_binding (-m gpu)
                                  binding.cpp with INTEGRATION.md 2's three edits (cross-built into
                                  oracle/_ref/binding_gpu/, which travels to the GPU box) under this repo's
                                  wrapper: gpu_rnnt of the reference binding on the MI355X library.
  test_reference_test_gpu_cu (-m gpu)
                                  /root/reference/tests/test_gpu.cu UNMODIFIED (oracle/build_ref_gpu_tests.sh maps its
                                  nine CUDA runtime names to HIP on the compiler command line) -> small_test,
                                  options_test, inf_test and the finite-difference grad_check of the reference's own GPU
                                  test program against this library on the MI355X: "Tests pass" (tests/test_gpu.cu:476-500)
  test_reference_test_time_cu (-m gpu)
                                  the reference's timing harness tests/test_time.cu, same build, one small run.
The first two need /root/reference (absent on the GPU box: skipped there)."""
import glob
import importlib.util
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIBDIR = os.path.join(ROOT, "warp-transducer_amd", "lib")
have_ref = os.path.exists(os.path.join(REF, "tests", "test_cpu.cpp"))


@pytest.mark.skipif(not have_ref, reason="/root/reference is not present")
def test_reference_test_cpu_cpp(tmp_path):
    src = open(os.path.join(REF, "tests", "test_cpu.cpp")).read()
    src, n = re.subn(r"^float numeric_grad", "void numeric_grad", src, flags=re.M)
    assert n == 1
    patched = tmp_path / "test_cpu.cpp"
    patched.write_text(src)
    exe = tmp_path / "test_cpu"
    # -I include: THIS repo's rnnt.h; the reference's include/ is not on the path (tests/test.h is)
    subprocess.run(["g++", "-O1", "-std=c++11", "-fopenmp", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(REF, "tests"), str(patched), os.path.join(REF, "tests", "random.cpp"),
                    "-o", str(exe), "-L" + LIBDIR, "-lwarprnnt", "-Wl,-rpath," + LIBDIR], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Tests pass" in out.stdout, out.stdout


@pytest.mark.skipif(not have_ref, reason="/root/reference is not present")
def test_reference_binding_and_test_py(tmp_path):
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "build_ref_binding.py"), "cpu"], check=True)
    mod = glob.glob(os.path.join(ROOT, "oracle", "_ref", "binding_cpu", "warp_rnnt*.so"))
    assert mod
    # the package layout the reference's setup.py installs: warprnnt_pytorch/{__init__.py, warp_rnnt*.so}
    pkg = tmp_path / "warprnnt_pytorch"
    pkg.mkdir()
    os.symlink(os.path.join(REF, "pytorch_binding", "warprnnt_pytorch", "__init__.py"), pkg / "__init__.py")
    os.symlink(mod[0], pkg / os.path.basename(mod[0]))
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    probe = subprocess.run([sys.executable, "-c", "import warprnnt_pytorch as w; print(w.__file__); "
                            "print(w.cpu_rnnt.__module__)"], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert probe.returncode == 0, probe.stderr
    assert os.path.realpath(probe.stdout.split()[0]).startswith(REF), probe.stdout    # the reference's wrapper, not ours
    out = subprocess.run([sys.executable, os.path.join(REF, "pytorch_binding", "test", "test.py")],
                         capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CPU Tests passed!" in out.stdout, out.stdout


def _load_gpu_binding():
    mod = glob.glob(os.path.join(ROOT, "oracle", "_ref", "binding_gpu", "warp_rnnt*.so"))
    if not mod:
        pytest.skip("oracle/_ref/binding_gpu was not built (oracle/build_ref_binding.py gpu)")
    spec = importlib.util.spec_from_file_location("warp_rnnt", mod[0])
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.gpu
def test_reference_gpu_binding(monkeypatch, oracle):
    """gpu_rnnt of the reference's binding.cpp (three edits) on the MI355X library: the
    golden vectors of pytorch_binding/test/test.py through this repo's wrapper with its extension module
    swapped for the reference-built one, and a random case against the oracle in fp32 and fp64."""
    import torch
    import warprnnt_pytorch
    from tests.golden import literals as G
    ref_mod = _load_gpu_binding()
    assert ref_mod.gpu_rnnt.__doc__ and "RNNT GPU version" in ref_mod.gpu_rnnt.__doc__   # binding.cpp:160
    monkeypatch.setattr(warprnnt_pytorch, "_ASYNC_GPU", False)         # the reference flow: gpu_rnnt in forward
    monkeypatch.setattr(warprnnt_pytorch, "warp_rnnt", ref_mod)
    dev = torch.device("cuda:0")
    for acts_np, labels, cost, grads_ref in ((G.SMALL_ACTS, [[1, 2]], G.SMALL_COST, G.SMALL_GRADS),
                                             (G.BIG_ACTS, [[1, 2], [1, 1]], sum(G.OPTIONS_COSTS), G.BIG_GRADS)):
        x = torch.tensor(acts_np, dtype=torch.float32, device=dev, requires_grad=True)
        n = x.shape[0]
        lab = torch.tensor(labels, dtype=torch.int32, device=dev)
        tl = torch.full((n,), x.shape[1], dtype=torch.int32, device=dev)
        ll = torch.tensor([len(l) for l in labels], dtype=torch.int32, device=dev)
        loss = warprnnt_pytorch.RNNTLoss(reduction="sum")(x, lab, tl, ll)
        loss.sum().backward()
        assert np.allclose(loss.item(), cost, rtol=1e-5)                              # test.py:75,155
        assert np.allclose(x.grad.cpu().numpy(), grads_ref, rtol=1e-3, atol=1e-6)     # test.py:77,158
    rng = np.random.default_rng(12)
    N, T, U, A = 3, 25, 7, 33
    acts = rng.standard_normal((N, T, U, A))
    labels = rng.integers(1, A, size=(N, U - 1)).astype(np.int32)
    tl, ll = np.array([T, 11, T], dtype=np.int32), np.array([U - 1, 2, 0], dtype=np.int32)
    for dtype, tol in ((torch.float32, 1e-4), (torch.float64, 1e-9)):
        x = torch.tensor(acts, dtype=dtype, device=dev)
        ref_c, ref_g = oracle.rnnt_logits(x.double().cpu().numpy(), labels, tl, ll)
        costs = torch.zeros(N, dtype=dtype)
        grads = torch.zeros_like(x)
        assert ref_mod.gpu_rnnt(x, torch.tensor(labels, device=dev), torch.tensor(tl, device=dev),
                                torch.tensor(ll, device=dev), costs, grads, 0, 0) == 0
        assert np.abs(costs.double().numpy() - ref_c).max() <= tol * max(1.0, np.abs(ref_c).max())
        assert np.abs(grads.double().cpu().numpy() - ref_g).max() <= tol


def _ref_binary(name):
    path = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s was not built (oracle/build_ref_gpu_tests.sh)" % name)
    return path


@pytest.mark.gpu
def test_reference_test_gpu_cu():
    out = subprocess.run([_ref_binary("ref_test_gpu")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Tests pass" in out.stdout, out.stdout[-2000:]
    for name in ("small_test 1", "options_test 1", "inf_test 1"):
        assert "finish " + name in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
def test_reference_test_time_cu():
    """tests/test_time.cu: `test_time B T L A` (README row T=150, L=40, A=28, N=16): ten timed calls, it prints their mean."""
    out = subprocess.run([_ref_binary("ref_test_time"), "16", "150", "40", "28"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "average 10 time cost" in out.stdout, out.stdout[-2000:]
