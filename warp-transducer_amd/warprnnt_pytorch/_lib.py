"""Loader for libwarprnnt.so (the C-ABI of include/rnnt.h) through ctypes.

This is the binding a maintainer of the reference would write instead of
pytorch_binding/src/binding.cpp: it passes raw device/host pointers, sizes and the
by-value ``rnntOptions`` straight to the exported C symbols.  There is NO fallback: if the
shared library is missing or lacks a symbol the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

RNNT_CPU, RNNT_GPU = 0, 1
STATUS_SUCCESS = 0
RNNT_STATUS_INVALID_VALUE, RNNT_STATUS_EXECUTION_FAILED = 2, 3          # include/rnnt.h:48-52

# dtype codes of compute_rnnt_loss_async (include/rnnt.h)
DT_F32, DT_F64, DT_BF16, DT_F16 = 0, 1, 2, 3


class rnntOptions(C.Structure):
    """include/rnnt.h `struct rnntOptions` (reference include/rnnt.h:43-64)."""
    _fields_ = [("loc", C.c_int), ("num_threads", C.c_uint), ("stream", C.c_void_p),
                ("blank_label", C.c_int), ("maxT", C.c_int), ("maxU", C.c_int),
                ("batch_first", C.c_bool)]


_PTR = C.c_void_p
_LOSS_ARGS = [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR, rnntOptions]

EXPORTS = {
    # name: (restype, argtypes)
    "get_warprnnt_version": (C.c_int, []),
    "rnntGetStatusString": (C.c_char_p, [C.c_int]),
    "compute_rnnt_loss": (C.c_int, _LOSS_ARGS),
    "compute_rnnt_loss_fp64": (C.c_int, _LOSS_ARGS),
    "get_workspace_size": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_bool,
                                     C.POINTER(C.c_size_t), C.c_size_t]),
    "compute_rnnt_loss_bf16": (C.c_int, _LOSS_ARGS),
    "compute_rnnt_loss_fp16": (C.c_int, _LOSS_ARGS),
    "compute_rnnt_loss_async": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR,
                                          _PTR, _PTR, rnntOptions, C.c_int]),
    "compute_rnnt_loss_sharded": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR, _PTR, _PTR, _PTR,
                                            rnntOptions, C.c_int]),
    "compute_rnnt_loss_fwd": (C.c_int, [_PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR, rnntOptions,
                                        C.c_int, C.c_int]),
    "compute_rnnt_loss_bwd": (C.c_int, [_PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, rnntOptions, C.c_int]),
    "compute_rnnt_loss_likelihoods": (C.c_int, [_PTR, C.c_int, rnntOptions, C.c_int, _PTR, _PTR]),
    "compute_rnnt_loss_lattice_dump": (C.c_int, [_PTR, _PTR, _PTR, C.c_int, C.c_int, rnntOptions, C.c_int, _PTR, _PTR]),
    "compute_rnnt_loss_fastemit": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR,
                                             _PTR, _PTR, rnntOptions, C.c_int, C.c_float]),
    "compute_rnnt_loss_fwd_fastemit": (C.c_int, [_PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR, rnntOptions,
                                                 C.c_int, C.c_int, C.c_float]),
    "compute_rnnt_loss_packed": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, _PTR, C.c_longlong, C.c_int, C.c_int, _PTR,
                                           _PTR, _PTR, rnntOptions, C.c_int, C.c_float]),
    "compute_rnnt_loss_packed_fwd": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_longlong, C.c_int, C.c_int, _PTR,
                                               _PTR, rnntOptions, C.c_int, C.c_int, C.c_float]),
    "compute_rnnt_loss_packed_bwd": (C.c_int, [_PTR, _PTR, _PTR, _PTR, C.c_longlong, C.c_int, C.c_int, _PTR,
                                               rnntOptions, C.c_int]),
    "get_workspace_size_add": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "compute_rnnt_loss_add": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR,
                                        rnntOptions]),
    "compute_rnnt_loss_add_fwd": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR, rnntOptions,
                                            C.c_int]),
    "compute_rnnt_loss_add_fwd_fastemit": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR,
                                                     rnntOptions, C.c_int, C.c_float]),
    "compute_rnnt_loss_add_bwd": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR,
                                            rnntOptions]),
    "compute_rnnt_loss_add_fwd_dt": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR, _PTR, rnntOptions,
                                               C.c_int, C.c_int, C.c_float]),
    "compute_rnnt_loss_add_bwd_dt": (C.c_int, [_PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _PTR, _PTR, C.c_int, C.c_int, _PTR,
                                               rnntOptions, C.c_int]),
    "get_warprnnt_extension_version": (C.c_int, []),
    "rnnt_host_staging": (C.c_int, [C.c_int]),
    "rnnt_host_staging_bytes": (C.c_longlong, []),
    "rnnt_host_staging_release": (C.c_longlong, []),
    "rnnt_set_rccl_all_reduce": (None, [_PTR]),
    "rnnt_sharded_prepare": (C.c_int, [_PTR]),
    "rnnt_sharded_release": (None, [_PTR]),
    "rnnt_set_aux_stream": (None, [_PTR]),
    "rnnt_rccl_source": (C.c_char_p, []),
    "rnnt_profile_enable": (None, [C.c_int]),
    "rnnt_profile_reset": (None, []),
    "rnnt_profile_collect": (None, []),
    "rnnt_profile_read": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
}

_lib = None


def library_path():
    """WARP_RNNT_PATH (a directory, as in the reference's pytorch_binding/setup.py:17) wins."""
    env = os.environ.get("WARP_RNNT_PATH")
    if env:
        return os.path.join(env, "libwarprnnt.so") if os.path.isdir(env) else env
    installed = os.path.join(_HERE, "lib", "libwarprnnt.so")          # pip-installed package: the library travels inside it
    if os.path.exists(installed):
        return installed
    return os.path.normpath(os.path.join(_HERE, "..", "lib", "libwarprnnt.so"))   # source tree: warp-transducer_amd/lib


def lib():
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                "libwarprnnt.so not found at %s -- build it with `make -C warp-transducer_amd` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no Python/CPU fallback for the HIP path." % path)
        handle = C.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing: intended
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def status_string(code):
    return lib().rnntGetStatusString(int(code)).decode()


def check(code, what):
    if code != STATUS_SUCCESS:
        raise RuntimeError("%s failed: %s (status %d)" % (what, status_string(code), code))


def workspace_bytes(maxT, maxU, minibatch, gpu, dtype_size):
    n = C.c_size_t(0)
    check(lib().get_workspace_size(int(maxT), int(maxU), int(minibatch), bool(gpu), C.byref(n),
                                   int(dtype_size)), "get_workspace_size")
    return n.value


def workspace_bytes_add(maxT, maxU, minibatch):
    """Workspace of the additive-joint entries (compute_rnnt_loss_add*)."""
    n = C.c_size_t(0)
    check(lib().get_workspace_size_add(int(maxT), int(maxU), int(minibatch), C.byref(n)), "get_workspace_size_add")
    return n.value
