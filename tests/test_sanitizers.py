"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md 5, row "race detection / sanitizers"; the
reference has no such target: /root/reference/CMakeLists.txt sets no sanitizer flag).

`make asan` (warp-transducer_amd/Makefile) builds lib/asan/libwarprnnt.so: the entry points of include/rnnt.h (argument
validation, workspace arithmetic), the RNNT_CPU location (csrc/rnnt_cpu.cpp) and the host driver, instrumented with
-fsanitize=address,undefined (device code is not: -fno-gpu-sanitize).  CPU only, no GPU needed:

  * the host suites (tests/test_cpu_location.py, tests/test_abi.py, tests/test_binding_cpu.py) run against that library --
    the ctypes loader takes it through WARP_RNNT_PATH, the sanitizer runtime is preloaded into the interpreter;
  * the reference's own tests/test_cpu.cpp (small_test, options_test, inf_test, grad_check), compiled with the same flags,
    runs against it -> "Tests pass" with no sanitizer report.

Found by the first runs (round 5), both fixed: rnntGetStatusString(99) and an unknown options.loc (7) loaded values outside
the enumerators' range through the enum types -- undefined behaviour in C++, and exactly what a C caller or ctypes can pass
(the reference has the same two loads: src/rnnt_entrypoint.cpp:18-35,61,77).  The entry points now read both as ints.
"""
import glob
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "warp-transducer_amd")
ASAN_DIR = os.path.join(PKG, "lib", "asan")
REF = "/root/reference"

SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86",
           "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=86"}


def clang_dir():
    out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    m = re.search(r"^InstalledDir: (.*)$", out, re.M)
    return m.group(1).strip() if m else None


def asan_runtime():
    d = clang_dir()
    if d is None:
        return None
    hits = glob.glob(os.path.join(os.path.dirname(d), "lib", "clang", "*", "lib", "linux", "libclang_rt.asan-x86_64.so"))
    return hits[0] if hits else None


@pytest.fixture(scope="module")
def asan_lib():
    if shutil.which("hipcc") is None or asan_runtime() is None:
        pytest.skip("needs the ROCm clang and its shared ASan runtime")
    out = subprocess.run(["make", "-j3", "-C", PKG, "asan"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lib = os.path.join(ASAN_DIR, "libwarprnnt.so")
    assert os.path.exists(lib)
    ldd = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "libclang_rt.asan" in ldd, ldd            # really instrumented (the runtime is a dependency)
    return lib


def test_host_suites_under_asan_and_ubsan(asan_lib):
    """tests/test_cpu_location.py, test_abi.py, test_binding_cpu.py with the instrumented library: green, and no sanitizer
    report (a report ends the process with exit code 86: halt_on_error)."""
    env = dict(os.environ, LD_PRELOAD=asan_runtime(), WARP_RNNT_PATH=ASAN_DIR, **SAN_ENV)
    env.pop("WARPRNNT_BINDING", None)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from warprnnt_pytorch import _lib, warp_rnnt; "
                            "print(_lib.library_path(), warp_rnnt.binding())" % PKG], capture_output=True, text=True, env=env, timeout=600)
    assert probe.returncode == 0, probe.stderr[-3000:]
    assert probe.stdout.split()[0] == asan_lib and probe.stdout.split()[1] == "ctypes", probe.stdout   # the instrumented library is the one under test
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_cpu_location.py"), os.path.join(ROOT, "tests", "test_abi.py"),
                          os.path.join(ROOT, "tests", "test_binding_cpu.py"),
                          # (asks for the compiled extension module, which is linked to the RELEASE library: with WARP_RNNT_PATH naming
                          #  another library the package switches to the ctypes loader, by design)
                          "--deselect", "tests/test_binding_cpu.py::test_compiled_module_exports_and_additive_joint_checks"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "runtime error" not in out.stdout + out.stderr and "AddressSanitizer" not in out.stdout + out.stderr, tail
    assert re.search(r"\b\d+ passed", out.stdout), tail


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "tests", "test_cpu.cpp")), reason="/root/reference is not present")
def test_reference_test_cpu_cpp_under_asan_and_ubsan(asan_lib, tmp_path):
    """The reference's own CPU test program (one token patched: `float numeric_grad` has no return statement, UB of the
    HARNESS), compiled with the same sanitizers, against the instrumented library."""
    src = open(os.path.join(REF, "tests", "test_cpu.cpp")).read()
    src, n = re.subn(r"^float numeric_grad", "void numeric_grad", src, flags=re.M)
    assert n == 1
    patched = tmp_path / "test_cpu.cpp"
    patched.write_text(src)
    exe = tmp_path / "test_cpu"
    clangxx = os.path.join(clang_dir(), "clang++")
    build = subprocess.run([clangxx, "-O1", "-g", "-std=c++11", "-fopenmp=libgomp", "-fsanitize=address,undefined", "-shared-libsan",
                            "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REF, "tests"),
                            str(patched), os.path.join(REF, "tests", "random.cpp"), "-o", str(exe), "-L" + ASAN_DIR, "-lwarprnnt",
                            "-Wl,-rpath," + ASAN_DIR, "-Wl,-rpath," + os.path.dirname(asan_runtime())], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=dict(os.environ, **SAN_ENV))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "Tests pass" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stdout + out.stderr[-2000:]
