"""pip-installable `warprnnt_pytorch` for the MI355X library (the role of the reference's pytorch_binding/setup.py:1-60).

    pip install . --no-build-isolation          # builds libwarprnnt.so with hipcc (gfx950) and the extension module with g++
    WARP_RNNT_PATH=/dir/with/libwarprnnt.so pip install . --no-build-isolation      # a prebuilt library, as the reference asks for

The installed package is self-contained: warprnnt_pytorch/{*.py, _warp_rnnt_ext*.so, lib/libwarprnnt.so, include/rnnt.h};
no sys.path edits, no environment variables at run time (a WARP_RNNT_PATH naming ANOTHER library at run time is honoured by switching to
the ctypes loader: the compiled module is linked to the library it was built with).
torch must be importable at build time (--no-build-isolation), exactly as for the reference's setup.py, which imports it.
"""
import importlib.util
import os
import shutil
import subprocess
import sys

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop
from setuptools.dist import Distribution

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_SRC = os.path.join(ROOT, "warp-transducer_amd", "warprnnt_pytorch")


class BinaryDistribution(Distribution):
    def has_ext_modules(self):          # platform wheel: the package carries two shared objects
        return True


class build_native(build_py):
    """Python sources, then the native pair into the package directory of the build tree."""

    def run(self):
        super().run()
        if getattr(self, "editable_mode", False):
            # PEP 660 (`pip install -e .` with setuptools >= 64, editable_wheel): the package is imported from the SOURCE tree, so
            # the native pair must be built there -- build_lib is a temporary directory nothing will ever import from
            _build_in_tree()
            return
        pkg_out = os.path.join(self.build_lib, "warprnnt_pytorch")
        lib_out = os.path.join(pkg_out, "lib")
        os.makedirs(lib_out, exist_ok=True)
        prebuilt = os.environ.get("WARP_RNNT_PATH")
        if prebuilt:
            src = os.path.join(prebuilt, "libwarprnnt.so") if os.path.isdir(prebuilt) else prebuilt
            if not os.path.exists(src):
                raise SystemExit("Could not find libwarprnnt.so in %s (WARP_RNNT_PATH)" % prebuilt)
        else:
            subprocess.run(["make", "-j3", "-C", os.path.join(ROOT, "warp-transducer_amd"), "lib/libwarprnnt.so"], check=True)
            src = os.path.join(ROOT, "warp-transducer_amd", "lib", "libwarprnnt.so")
        shutil.copy2(src, os.path.join(lib_out, "libwarprnnt.so"))
        inc_out = os.path.join(pkg_out, "include")
        os.makedirs(inc_out, exist_ok=True)
        shutil.copy2(os.path.join(ROOT, "include", "rnnt.h"), os.path.join(inc_out, "rnnt.h"))
        spec = importlib.util.spec_from_file_location("_warprnnt_build_ext", os.path.join(PKG_SRC, "build_ext.py"))
        be = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(be)
        be.build(force=True, lib_dir=lib_out, out_dir=pkg_out)


def _build_in_tree():
    if not os.environ.get("WARP_RNNT_PATH"):
        subprocess.run(["make", "-j3", "-C", os.path.join(ROOT, "warp-transducer_amd"), "lib/libwarprnnt.so"], check=True)
    subprocess.run([sys.executable, os.path.join(PKG_SRC, "build_ext.py")], check=True)


class develop_native(develop):
    """`pip install -e .` / `setup.py develop`: the package runs from the source tree, so the native pair is built IN the
    tree (what __graft_entry__.build() does) -- an editable install without it would silently fall back to the ctypes
    loader, or find no library at all."""

    def run(self):
        _build_in_tree()
        super().run()


setup(
    name="warprnnt_pytorch",
    version="0.4.0",
    description="RNN-Transducer loss for AMD MI355X (gfx950): drop-in for HawkAaron/warp-transducer's PyTorch binding",
    packages=["warprnnt_pytorch"],
    package_dir={"warprnnt_pytorch": os.path.relpath(PKG_SRC, ROOT)},
    python_requires=">=3.8",
    cmdclass={"build_py": build_native, "develop": develop_native},
    distclass=BinaryDistribution,
    zip_safe=False,
)
