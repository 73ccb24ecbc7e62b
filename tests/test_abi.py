"""The C-ABI shared library: loads, exports every symbol include/rnnt.h declares, and keeps the
reference's validation / status behaviour (src/rnnt_entrypoint.cpp:14-35,49-59,96-128).
No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from warprnnt_pytorch import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rnnt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", re.sub(r"#.*", "", src)))
                  - {"defined", "sizeof"})


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = declared_functions()
    assert {"compute_rnnt_loss", "compute_rnnt_loss_fp64", "get_workspace_size", "get_warprnnt_version",
            "rnntGetStatusString"} <= set(names)
    for n in names:
        assert hasattr(lib, n), "libwarprnnt.so does not export %s" % n
    assert set(names) == set(_lib.EXPORTS), (names, sorted(_lib.EXPORTS))


def test_dynamic_symbol_table_is_the_c_abi_only():
    """The reference's libwarprnnt.so exports its five C entry points and nothing else; this one exports
    exactly what include/rnnt.h declares (no C++ helpers, no host-side kernel handles)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.library_path()], check=True, capture_output=True,
                         text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == set(declared_functions()), sorted(exported ^ set(declared_functions()))


def test_version_and_status_strings():
    lib = _lib.lib()
    assert lib.get_warprnnt_version() == 1                       # tests/test_cpu.cpp:382
    assert [_lib.status_string(i) for i in range(5)] == [
        "no error", "cuda memcpy or memset failed", "invalid value", "execution failed", "unknown error"]
    assert _lib.status_string(99) == "unknown error"


def test_options_struct_layout():
    # by-value struct of the reference: 32 bytes on LP64
    assert C.sizeof(_lib.rnntOptions) == 32
    assert _lib.rnntOptions.stream.offset == 8 and _lib.rnntOptions.batch_first.offset == 28


def test_workspace_size_contract():
    lib = _lib.lib()
    n = C.c_size_t(0)
    for bad in ((0, 3, 1), (4, 0, 1), (4, 3, 0), (-1, 3, 1)):
        assert lib.get_workspace_size(*bad, True, C.byref(n), 4) == 2        # INVALID_VALUE
    assert lib.get_workspace_size(4, 3, 2, False, C.byref(n), 4) == 0
    assert n.value == 4 * 2 * 4 * 3 * 4                                      # 4*N*T*U*s (CPU location)
    assert lib.get_workspace_size(150, 21, 128, True, C.byref(n), 4) == 0
    gpu32 = n.value
    assert lib.get_workspace_size(150, 21, 128, True, C.byref(n), 2) == 0
    assert n.value == gpu32                                                  # 16-bit acts keep an fp32 lattice
    assert lib.get_workspace_size(150, 21, 128, True, C.byref(n), 8) == 0
    assert n.value > gpu32
    assert lib.get_workspace_size_add(150, 21, 128, C.byref(n)) == 0         # additive joint: + maxima and W planes
    assert n.value > gpu32
    assert lib.get_workspace_size_add(0, 21, 128, C.byref(n)) == 2
    # must hold the skewed lattice: 5 fp32 words per cell of N*(T+U-1)*U
    assert gpu32 >= 5 * 4 * 128 * (150 + 21 - 1) * 21


@pytest.mark.parametrize("fn", ["compute_rnnt_loss", "compute_rnnt_loss_fp64"])
def test_invalid_value_paths(fn):
    lib = _lib.lib()
    f = getattr(lib, fn)
    x = np.zeros(64, dtype=np.float64)
    i = np.ones(8, dtype=np.int32)
    ok = dict(acts=x.ctypes.data, grads=None, lab=i.ctypes.data, ll=i.ctypes.data, tl=i.ctypes.data,
              A=3, N=1, costs=x.ctypes.data, ws=x.ctypes.data)

    def call(opt, **over):
        a = dict(ok, **over)
        return f(a["acts"], a["grads"], a["lab"], a["ll"], a["tl"], a["A"], a["N"], a["costs"], a["ws"], opt)

    good = _lib.rnntOptions(loc=0, num_threads=1, stream=None, blank_label=0, maxT=1, maxU=2, batch_first=True)
    for key in ("acts", "lab", "ll", "tl", "costs", "ws"):
        assert call(good, **{key: None}) == 2
    assert call(good, A=0) == 2 and call(good, N=0) == 2
    for bad in (dict(maxT=0), dict(maxU=0), dict(loc=7)):
        kw = dict(loc=0, num_threads=1, stream=None, blank_label=0, maxT=1, maxU=2, batch_first=True)
        kw.update(bad)
        assert call(_lib.rnntOptions(**kw)) == 2
    # extensions refuse the CPU location
    if fn == "compute_rnnt_loss":
        assert lib.compute_rnnt_loss_bf16(ok["acts"], None, ok["lab"], ok["ll"], ok["tl"], 3, 1, ok["costs"],
                                          ok["ws"], good) == 2


def test_loader_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setenv("WARP_RNNT_PATH", str(tmp_path))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(ImportError):
        _lib.lib()


@pytest.mark.parametrize("lang,std,cc", [("c", "c99", "gcc"), ("c++", "c++11", "g++")])
def test_header_is_plain_c_and_cxx(tmp_path, lang, std, cc):
    """include/rnnt.h is the drop-in boundary: a C99 or C++11 translation unit that only includes it must
    compile (no torch / HIP types in the signatures) and see every entry point as a C symbol."""
    import shutil
    import subprocess
    if shutil.which(cc) is None:
        pytest.skip("%s not installed" % cc)
    src = tmp_path / ("use_rnnt." + ("c" if lang == "c" else "cpp"))
    calls = "\n".join("    p = (void*)&%s; (void)p;" % n for n in declared_functions())
    src.write_text('#include "rnnt.h"\nint main(void) {\n    void* p;\n%s\n    return 0;\n}\n' % calls)
    r = subprocess.run([cc, "-std=" + std, "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr



def test_round3_extension_entries_without_a_gpu():
    """What the round-3 extension entries do before they touch a device: the extension revision, the staging switch (off
    unless asked for; nothing allocated by toggling it), and INVALID_VALUE for missing arguments / the CPU location."""
    lib = _lib.lib()
    assert lib.get_warprnnt_extension_version() == 5          # 4: lattice dump; 5: rnnt_sharded_prepare, in-place gradients
    # a communicator has to be introduced before a sharded step may name it: NULL is refused, an unprepared one is INVALID_VALUE
    assert lib.rnnt_sharded_prepare(None) == 2
    lib.rnnt_sharded_release(C.c_void_p(0x1234))                # (unknown: ignored)
    assert lib.rnnt_host_staging(-1) == 0 and lib.rnnt_host_staging_bytes() == 0        # default: the library allocates nothing
    assert lib.rnnt_host_staging(1) == 0 and lib.rnnt_host_staging(-1) == 1
    assert lib.rnnt_host_staging(0) == 1 and lib.rnnt_host_staging(-1) == 0
    assert lib.rnnt_host_staging_bytes() == 0 and lib.rnnt_host_staging_release() == 0
    x = np.zeros(64, dtype=np.float64)
    i = np.ones(8, dtype=np.int32)
    gpu = _lib.rnntOptions(loc=1, num_threads=0, stream=None, blank_label=0, maxT=1, maxU=2, batch_first=True)
    cpu = _lib.rnntOptions(loc=0, num_threads=1, stream=None, blank_label=0, maxT=1, maxU=2, batch_first=True)
    p = x.ctypes.data
    # likelihood read-back: NULL outputs, bad dtype code, CPU location
    assert lib.compute_rnnt_loss_likelihoods(p, 1, gpu, 0, None, p) == 2
    assert lib.compute_rnnt_loss_likelihoods(p, 1, gpu, 7, p, p) == 2
    assert lib.compute_rnnt_loss_likelihoods(p, 1, cpu, 0, p, p) == 2
    assert lib.compute_rnnt_loss_likelihoods(None, 1, gpu, 0, p, p) == 2
    # sharded step: NULL pair, CPU location, NULL activations
    args = (p, None, i.ctypes.data, i.ctypes.data, i.ctypes.data, 3, 1, p, None)
    assert lib.compute_rnnt_loss_sharded(*args, None, None, p, gpu, 0) == 2
    assert lib.compute_rnnt_loss_sharded(*args, p, None, p, cpu, 0) == 2
    assert lib.compute_rnnt_loss_sharded(*args, p, C.c_void_p(0x1234), p, gpu, 0) == 2      # a communicator nobody prepared
    assert lib.compute_rnnt_loss_sharded(None, None, i.ctypes.data, i.ctypes.data, i.ctypes.data, 3, 1, p, None, p, None, p, gpu, 0) == 2


def test_workspace_size_is_monotone_in_every_argument():
    """get_workspace_size / get_workspace_size_add never shrink when a dimension, the batch or the element size grows (callers that
    cache a workspace for "the largest shape so far" rely on it), for both locations; the GPU formula is private but
    at least the reference's (3 T U + 2) N s (src/rnnt_entrypoint.cpp:96-128), so a buffer sized for this library also serves it."""
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    import ctypes as C

    def size(T, U, N, gpu, esz):
        return _lib.workspace_bytes(T, U, N, gpu, esz)

    def size_add(T, U, N):
        n = C.c_size_t()
        assert lib.get_workspace_size_add(T, U, N, C.byref(n)) == 0
        return n.value

    Ts, Us, Ns = [1, 2, 7, 16, 33, 150, 151, 1500], [1, 2, 8, 9, 21, 41, 64, 65, 301, 1024], [1, 2, 3, 16, 64, 128, 1024]
    for gpu in (True, False):
        for esz in (2, 4, 8):
            for U in Us:
                for N in (1, 16):
                    vals = [size(T, U, N, gpu, esz) for T in Ts]
                    assert vals == sorted(vals), ("T", gpu, esz, U, N, vals)
            for T in (1, 33, 150):
                for N in (1, 16):
                    vals = [size(T, U, N, gpu, esz) for U in Us]
                    assert vals == sorted(vals), ("U", gpu, esz, T, N, vals)
                for U in (1, 21, 65):
                    vals = [size(T, U, N, gpu, esz) for N in Ns]
                    assert vals == sorted(vals), ("N", gpu, esz, T, U, vals)
        for T, U, N in ((150, 21, 128), (1500, 301, 64), (7, 3, 2)):
            by_esz = [size(T, U, N, gpu, e) for e in (2, 4, 8)]
            assert by_esz == sorted(by_esz)
            if gpu:
                assert size(T, U, N, True, 4) >= (3 * T * U + 2) * N * 4
                assert size_add(T, U, N) >= size(T, U, N, True, 4)            # a workspace of the additive-joint size serves every GPU entry
    vals = [size_add(T, 21, 16) for T in Ts]
    assert vals == sorted(vals)
    vals = [size_add(150, U, 16) for U in Us]
    assert vals == sorted(vals)
