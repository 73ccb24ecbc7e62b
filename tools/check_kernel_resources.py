#!/usr/bin/env python
"""No kernel of the library may use scratch memory or spill vector registers.

Scratch traffic is ruinous for these kernels (round 4: a gradient kernel with 4 spilled registers ran at a quarter of its
speed; the split-contraction DF kernel at four columns per lane needed > 512 registers and took 331 us where the two-column
form takes 194) and nothing in the test suite notices it -- results stay correct.  This script compiles the device code of both
translation units (hipcc --cuda-device-only -S), reads every kernel's resource record and fails if one has a private segment
or spilled VGPRs.  SGPR spills (into VGPR lanes: no memory traffic) are listed, not failed.
Usage: python tools/check_kernel_resources.py [-v]        (tests/test_isa_guard.py runs it)"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_lattice_lin_isa as guard                     # device_asm(): the same compile command


def kernels(asm):
    """[(demangled-ish name, dict of resource fields)] from the amdhsa.kernels metadata of one translation unit."""
    meta = asm[asm.index("amdhsa.kernels"):]
    out = []
    for block in meta.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda key: int((re.search(r"\.%s:\s+(\d+)" % key, block) or [None, "0"])[1])
        out.append((name, {"vgpr": get("vgpr_count"), "sgpr": get("sgpr_count"), "scratch": get("private_segment_fixed_size"),
                           "vspill": get("vgpr_spill_count"), "sspill": get("sgpr_spill_count"), "lds": get("group_segment_fixed_size")}))
    return out


def check(asm):
    """Problems (strings) of one translation unit's kernels."""
    probs = []
    for name, r in kernels(asm):
        if r["scratch"] or r["vspill"]:
            probs.append("%s: private segment %d bytes, %d vector registers spilled" % (name, r["scratch"], r["vspill"]))
    return probs


if __name__ == "__main__":
    verbose = "-v" in sys.argv
    bad = []
    for src in guard.SRCS:
        asm = guard.device_asm(None, src)
        ks = kernels(asm)
        bad += [os.path.basename(src) + ": " + p for p in check(asm)]
        if os.path.basename(src) not in guard.NO_LIN:
            bad += [os.path.basename(src) + ": lattice_lin_kernel: " + p for p in guard.check(asm)]     # (the hand-counted vmcnt, same compile)
        sp = [(n, r["sspill"]) for n, r in ks if r["sspill"]]
        print("%s: %d kernels, %d with SGPR spills (into VGPR lanes), most registers %d" % (os.path.basename(src), len(ks), len(sp), max(r["vgpr"] for _, r in ks)))
        if verbose:
            for n, r in ks:
                print("   %-110s %s" % (subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:110], r))
    if bad:
        print("device-code checks FAILED:")
        for p in bad:
            print("  *", p)
        sys.exit(1)
    print("no kernel uses scratch memory or spills vector registers; lattice_lin_kernel's operand role holds exactly its hand-written "
          "vector-memory instructions (both translation units)")
