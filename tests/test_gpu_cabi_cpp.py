"""-m gpu: the C++ consumers of the C-ABI (warp-transducer_amd/cabi_tests) -- the reference's
tests/test_gpu.cu golden tests and its tests/test_time.cu timing CLI, compiled against
include/rnnt.h and linked to libwarprnnt.so with no Python in between."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
BUILD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "warp-transducer_amd", "build")


def _binary(name):
    path = os.path.join(BUILD, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.dirname(BUILD), "cabi_tests"], check=True, stdout=subprocess.DEVNULL)
    return path


def test_cpp_golden_tests():
    out = subprocess.run([_binary("test_gpu")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Tests pass" in out.stdout
    for name in ("small_test", "options_test", "inf_test", "grad_check", "packed_test"):
        assert "finish %s 1" % name in out.stdout


def test_cpp_test_time_cli():
    # the README table's first configuration: T=150 L=40 A=28 N=16 (11.43 ms on a GTX 1080 Ti)
    out = subprocess.run([_binary("test_time"), "16", "150", "40", "28"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"average 10 time cost: ([0-9.]+) ms", out.stdout)
    assert m and float(m.group(1)) < 11.43            # includes the un-warmed first call, as the reference's protocol
