"""The boundary ships the way the reference ships it (pytorch_binding/setup.py:1-60, CMakeLists.txt:1-136):
`pip install .` gives an importable, self-contained `warprnnt_pytorch` (no sys.path edits, no environment at run time),
and `cmake -B b && cmake --build b` gives the same libwarprnnt.so as the Makefile (same export list)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.join(ROOT, "warp-transducer_amd", "lib")

CHECK = r"""
import os, sys
assert not any(p.rstrip('/').endswith(('repo', 'warp-transducer_amd')) for p in sys.path), sys.path
import torch, warprnnt_pytorch
from warprnnt_pytorch import RNNTLoss, warp_rnnt, _lib
here = os.path.dirname(warprnnt_pytorch.__file__)
assert here.startswith(sys.prefix), (here, sys.prefix)                     # the INSTALLED package, not the source tree
assert warp_rnnt.binding() == "ext"                                        # the compiled module came along ...
assert _lib.library_path() == os.path.join(here, "lib", "libwarprnnt.so")  # ... and so did the library
assert os.path.exists(os.path.join(here, "include", "rnnt.h"))
acts = torch.tensor([[[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
                      [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]], requires_grad=True)
loss = RNNTLoss(reduction='sum')(acts, torch.IntTensor([[1, 2]]), torch.IntTensor([2]), torch.IntTensor([2]))
loss.backward()
assert abs(loss.item() - 4.495666) < 1e-5, loss.item()                     # tests/test_cpu.cpp:26
assert abs(acts.grad[0, 0, 0, 1].item() + 0.3999269) < 1e-6
print("installed-ok", _lib.lib().get_warprnnt_version())
"""


def test_pip_install_into_a_clean_venv(tmp_path):
    """`pip install .` (with WARP_RNNT_PATH naming the built library, as the reference's setup.py expects; without it
    setup.py runs the hipcc build itself) into a fresh virtual environment, then import and use the package from a
    directory that is not the source tree."""
    if not os.path.exists(os.path.join(LIB_DIR, "libwarprnnt.so")):
        pytest.skip("libwarprnnt.so not built")
    venv = tmp_path / "venv"
    subprocess.run([sys.executable, "-m", "venv", "--without-pip", "--system-site-packages", str(venv)], check=True)   # (torch comes from the system)
    py = str(venv / "bin" / "python")
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "WARPRNNT_BINDING")}
    env["WARP_RNNT_PATH"] = LIB_DIR
    env["PIP_DISABLE_PIP_VERSION_CHECK"] = "1"
    work = tmp_path / "src"                       # (a copy: the build must not leave build/ and *.egg-info in the checkout)
    shutil.copytree(ROOT, work, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "build", "*.o", "__pycache__", "_ref",
                                                             ".pytest_cache", "*.egg-info", "dev"))
    out = subprocess.run([py, "-m", "pip", "install", str(work), "--no-build-isolation", "--no-deps", "-q"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    env.pop("WARP_RNNT_PATH")
    run = subprocess.run([py, "-c", CHECK], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "installed-ok 1" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]


def test_sdist_carries_the_native_sources(tmp_path):
    """An sdist must be buildable: the HIP / C++ sources, the public header, the Makefile and the extension module's source
    travel with it (MANIFEST.in)."""
    import tarfile
    work = tmp_path / "src"
    shutil.copytree(ROOT, work, ignore=shutil.ignore_patterns(".git", "gpurun_out", "profiles", "build", "*.o", "*.so", "__pycache__", "_ref",
                                                             ".pytest_cache", "*.egg-info", "dev", "oracle", "tests", "tools"))
    out = subprocess.run([sys.executable, "setup.py", "-q", "sdist", "-d", str(tmp_path / "dist")], cwd=str(work), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    (tarball,) = list((tmp_path / "dist").glob("*.tar.gz"))
    names = {n.split("/", 1)[1] for n in tarfile.open(tarball).getnames() if "/" in n}
    for need in ("include/rnnt.h", "warp-transducer_amd/Makefile", "warp-transducer_amd/exports.map", "warp-transducer_amd/csrc/rnnt_gpu.hip",
                 "warp-transducer_amd/csrc/rnnt_joint.hip", "warp-transducer_amd/csrc/rnnt_kernels.h", "warp-transducer_amd/csrc/rnnt_cpu.cpp",
                 "warp-transducer_amd/warprnnt_pytorch/csrc/binding.cpp", "warp-transducer_amd/warprnnt_pytorch/build_ext.py",
                 "warp-transducer_amd/warprnnt_pytorch/__init__.py", "setup.py", "pyproject.toml"):
        assert need in names, (need, sorted(names)[:40])
    assert not any(n.endswith((".so", ".o")) for n in names)


def test_cmake_build_matches_the_makefile_build(tmp_path):
    """cmake -B b && cmake --build b: libwarprnnt.so with exactly the export list of include/rnnt.h (= the Makefile build's),
    the two C-ABI consumers, and an install tree another CMake project finds with find_package(warprnnt)."""
    cmake, ninja = shutil.which("cmake"), shutil.which("ninja")
    if cmake is None or shutil.which("hipcc") is None:
        pytest.skip("needs cmake and the ROCm compiler")
    from tests.test_abi import declared_functions
    b = tmp_path / "b"
    gen = ["-G", "Ninja"] if ninja else []
    out = subprocess.run([cmake, "-S", ROOT, "-B", str(b)] + gen, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    out = subprocess.run([cmake, "--build", str(b), "-j", "3"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lib = b / "libwarprnnt.so"
    assert lib.exists() and (b / "test_gpu").exists() and (b / "test_time").exists()
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.run([nm, "-D", "--defined-only", str(lib)], check=True, capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in syms.splitlines() if line.strip()}
    assert exported == set(declared_functions()), sorted(exported ^ set(declared_functions()))
    mk = os.path.join(LIB_DIR, "libwarprnnt.so")
    if os.path.exists(mk):
        syms2 = subprocess.run([nm, "-D", "--defined-only", mk], check=True, capture_output=True, text=True).stdout
        assert exported == {line.split()[-1] for line in syms2.splitlines() if line.strip()}
    # install + find_package from a second project (host-only consumer: the version call needs no GPU)
    prefix = tmp_path / "prefix"
    out = subprocess.run([cmake, "--install", str(b), "--prefix", str(prefix)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert (prefix / "include" / "rnnt.h").exists() and (prefix / "lib" / "libwarprnnt.so").exists()
    user = tmp_path / "user"
    user.mkdir()
    (user / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.21)\nproject(user LANGUAGES CXX)\nfind_package(warprnnt REQUIRED)\n"
        "add_executable(v v.cpp)\ntarget_link_libraries(v PRIVATE warprnnt::warprnnt)\n")
    (user / "v.cpp").write_text('#include <rnnt.h>\n#include <cstdio>\nint main() { std::printf("%d %s\\n", get_warprnnt_version(), '
                                'rnntGetStatusString(RNNT_STATUS_INVALID_VALUE)); return 0; }\n')
    out = subprocess.run([cmake, "-S", str(user), "-B", str(user / "b"), "-DCMAKE_PREFIX_PATH=%s" % prefix] + gen, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    out = subprocess.run([cmake, "--build", str(user / "b")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    run = subprocess.run([str(user / "b" / "v")], capture_output=True, text=True, timeout=60,
                         env=dict(os.environ, LD_LIBRARY_PATH="%s:%s" % (prefix / "lib", os.environ.get("LD_LIBRARY_PATH", ""))))
    assert run.returncode == 0 and run.stdout.split()[0] == "1" and "invalid value" in run.stdout, run.stdout + run.stderr
