"""Builds `warprnnt_pytorch/_warp_rnnt_ext*.so`, the compiled PyTorch extension module (csrc/binding.cpp), IN TREE.

    python warp-transducer_amd/warprnnt_pytorch/build_ext.py [--force]

Host code only: one g++ command against the torch headers and libraries of the running interpreter, linked to
../lib/libwarprnnt.so (built first by `make -C warp-transducer_amd`) with an $ORIGIN-relative rpath, so the pair travels
together.  (The reference builds its binding with setuptools' CppExtension / CUDAExtension keyed on WARP_RNNT_PATH,
pytorch_binding/setup.py:1-60; setup.py at the repository root does the same for this library and calls into here.)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
NAME = "_warp_rnnt_ext"


def target(out_dir=HERE):
    return os.path.join(out_dir, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def command(lib_dir, out_dir=HERE, include_dir=None):
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH") or ce.ROCM_HOME or "/opt/rocm"
    inc = [include_dir or os.path.join(ROOT, "include"), sysconfig.get_paths()["include"], os.path.join(rocm, "include")]
    inc += ce.include_paths()
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi())]
    cmd += ["-I" + d for d in inc]
    cmd += [os.path.join(HERE, "csrc", "binding.cpp"), "-o", target(out_dir)]
    rel = os.path.relpath(os.path.abspath(lib_dir), os.path.abspath(out_dir))
    cmd += ["-L" + lib_dir, "-lwarprnnt", "-Wl,-rpath,$ORIGIN/" + rel, "-Wl,-rpath,$ORIGIN/lib",    # (source tree | installed package)
            "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-Wl,-rpath," + tlib,
            "-L" + os.path.join(rocm, "lib"), "-lamdhip64"]
    return cmd


def build(force=False, lib_dir=None, out_dir=HERE, quiet=False):
    lib_dir = lib_dir or os.path.join(PKG, "lib")
    lib = os.path.join(lib_dir, "libwarprnnt.so")
    if not os.path.exists(lib):
        raise RuntimeError("build libwarprnnt.so first (make -C warp-transducer_amd): %s not found" % lib)
    out = target(out_dir)
    srcs = [os.path.join(HERE, "csrc", "binding.cpp"), os.path.join(ROOT, "include", "rnnt.h"), lib]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs if os.path.exists(s)):
        return out
    cmd = command(lib_dir, out_dir)
    if not quiet:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
