#!/usr/bin/env python
"""Builds the REFERENCE's PyTorch extension module (pytorch_binding/src/binding.cpp) against THIS repo's
include/rnnt.h and libwarprnnt.so -- test infrastructure for the drop-in claim of INTEGRATION.md 2.

  oracle/_ref/binding_cpu/warp_rnnt*.so   binding.cpp exactly as it lies under /root/reference (CPU half)
  oracle/_ref/binding_gpu/warp_rnnt*.so   binding.cpp with the three edits of INTEGRATION.md 2 applied IN MEMORY
                                          (-DWARPRNNT_ENABLE_GPU): THC -> c10 hip headers (binding.cpp:7-10),
                                          the current HIP stream (binding.cpp:104), hipSetDevice + the caching
                                          allocator and the missing sizeof(double) (binding.cpp:115-148)
Both are extension modules called `warp_rnnt`, the name the reference's setup.py gives them
(`warprnnt_pytorch.warp_rnnt`), so the reference's own warprnnt_pytorch/__init__.py imports them unchanged.

No reference source is copied into the repository: the patched text only exists in a temporary directory
during the compile; the built modules land in the git-ignored oracle/_ref/ and travel to the GPU box like
libwarprnnt_ref.so (the GPU half is cross-built here: it is host code only, no kernels).
Run by oracle/Makefile when /root/reference is present.  Usage: build_ref_binding.py [cpu|gpu|all]"""
import os
import re
import shutil
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REF", "/root/reference")
SRC = os.path.join(REF, "pytorch_binding", "src", "binding.cpp")
OUT = os.path.join(HERE, "_ref")
LIBDIR = os.path.join(ROOT, "warp-transducer_amd", "lib")


def patched_for_hip(text):
    """the three edits a maintainer makes to build gpu_rnnt on ROCm PyTorch."""
    n = {}
    text, n["thc"] = re.subn(r'#include "THC.h"\s*\n\s*extern THCState\* state;',
                             '#include <hip/hip_runtime.h>\n    #include <c10/hip/HIPStream.h>\n'
                             '    #include <c10/hip/HIPCachingAllocator.h>', text)
    text, n["stream"] = re.subn(r"at::cuda::getCurrentCUDAStream\(\)",
                                "reinterpret_cast<CUstream>(c10::hip::getCurrentHIPStream().stream())", text)
    text, n["dev"] = re.subn(r"cudaSetDevice\(", "hipSetDevice(", text)
    text, n["malloc"] = re.subn(r"THCudaMalloc\(state, ", "c10::hip::HIPCachingAllocator::raw_alloc(", text)
    text, n["free"] = re.subn(r"THCudaFree\(state, ", "c10::hip::HIPCachingAllocator::raw_delete(", text)
    # the fp64 GPU case sizes its workspace with the fp32 default (binding.cpp:134-135)
    head, gpu = text.split("int gpu_rnnt(")
    parts = gpu.split("case torch::ScalarType::Double:")
    parts[1], n["f64"] = re.subn(r"true, &gpu_size_bytes\);", "true, &gpu_size_bytes, sizeof(double));", parts[1], count=1)
    text = head + "int gpu_rnnt(" + "case torch::ScalarType::Double:".join(parts)
    assert n == {"thc": 1, "stream": 1, "dev": 2, "malloc": 2, "free": 2, "f64": 1}, n
    return text


def build(kind):
    from torch.utils import cpp_extension
    name = "warp_rnnt"
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    target = os.path.join(OUT, "binding_%s" % kind, name + suffix)
    deps = [SRC, os.path.join(ROOT, "include", "rnnt.h"), os.path.abspath(__file__)]
    if os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps):
        print("up to date:", target)
        return target
    os.makedirs(os.path.dirname(target), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="refbind_")
    try:
        src = SRC
        cflags = ["-O1", "-I" + os.path.join(ROOT, "include")]
        ldflags = ["-L" + LIBDIR, "-lwarprnnt", "-Wl,-rpath," + LIBDIR]
        if kind == "gpu":
            src = os.path.join(tmp, "binding_hip.cpp")
            with open(src, "w") as f:
                f.write(patched_for_hip(open(SRC).read()))
            torch_lib = os.path.join(os.path.dirname(cpp_extension.__file__), "..", "lib")
            cflags += ["-DWARPRNNT_ENABLE_GPU", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-I/opt/rocm/include"]
            ldflags += ["-L" + os.path.normpath(torch_lib), "-lc10_hip", "-L/opt/rocm/lib", "-lamdhip64"]
        cpp_extension.load(name=name, sources=[src], extra_cflags=cflags, extra_ldflags=ldflags,
                           build_directory=tmp, is_python_module=False, verbose=False)
        built = [f for f in os.listdir(tmp) if f.startswith(name) and f.endswith(".so")]
        assert built, os.listdir(tmp)
        shutil.copy(os.path.join(tmp, built[0]), target)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("built", target)
    return target


if __name__ == "__main__":
    if not os.path.exists(SRC):
        print("reference checkout absent: keeping prebuilt oracle/_ref/binding_* (if any)")
        sys.exit(0)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "all":
        # one process per module: torch's JIT builder renames the second module of the same name it builds in a
        # process (warp_rnnt_v1), and both halves must be called warp_rnnt
        import subprocess
        for k in ("cpu", "gpu"):
            subprocess.run([sys.executable, os.path.abspath(__file__), k], check=True)
    else:
        build(what)
