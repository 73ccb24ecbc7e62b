"""The hand-counted vmcnt of lattice_lin_kernel's operand wavefronts (inline-asm buffer loads + s_waitcnt vmcnt(28)) is only
right while the compiler adds no vector-memory instruction of its own to that role.  tools/check_lattice_lin_isa.py reads the
device ISA of THIS toolchain's build and checks exactly that; here it must hold for the shipped source, and it must trip on
ISA that has such an instruction planted (so a compiler upgrade cannot break the count silently)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def device_asm():
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    import check_lattice_lin_isa as guard
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as pool:                       # the four translation units at once (hipcc, 15 ... 75 s each)
        jobs = [pool.submit(guard.device_asm, None, src) for src in guard.SRCS]
        asms = [j.result() for j in jobs]
    guard.joint_asm = asms[1]
    guard.other_asms = dict(zip([os.path.basename(p) for p in guard.SRCS[2:]], asms[2:]))      # fp64 and 16-bit storage
    return guard, asms[0]


def test_operand_role_holds_only_its_own_memory_instructions(device_asm):
    guard, asm = device_asm
    assert guard.check(asm) == []


def test_guard_trips_on_a_planted_instruction(device_asm):
    guard, asm = device_asm
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if (l.startswith("_ZN4rnnt18lattice_lin_kernel") or l.startswith("_ZN4rnntL18lattice_lin_kernel")) and ":" in l)
    waits = [i for i in range(start, len(lines)) if "vmcnt(%d)" % ((guard.PFD - 1) * guard.KW) in lines[i]]
    for where, what in ((waits[2] + 5, "\tglobal_load_dword v99, v98, s[0:1]"), (waits[11] + 3, "\tscratch_load_dword v7, off, s32")):
        bad = list(lines)
        bad.insert(where, what)                       # what a spill reload or a re-materialised load would look like
        probs = guard.check("\n".join(bad))
        assert probs and any("compiler-issued" in p for p in probs), probs
    # and on a changed request count (a ring refill the compiler dropped or duplicated)
    req = next(i for i in range(waits[0], waits[1]) if "buffer_load_dwordx2" in lines[i])
    bad = list(lines)
    del bad[req]
    assert guard.check("\n".join(bad))


def test_no_kernel_uses_scratch_or_spills_vector_registers(device_asm):
    """Every kernel of EVERY translation unit: no private segment, no spilled VGPR (tools/check_kernel_resources.py).  Scratch
    does not change results -- no parity test would notice -- and it cost the split-contraction DF kernel 40 % (EXPERIMENTS 11)."""
    guard, asm = device_asm
    import check_kernel_resources as res
    assert len(res.kernels(asm)) > 25 and res.check(asm) == []
    for name, other in guard.other_asms.items():          # the materialised path's code objects for fp64 / 16-bit storage
        assert len(res.kernels(other)) > 20 and res.check(other) == [], name
        if name not in guard.NO_LIN:
            assert guard.check(other) == [], name          # their copy of lattice_lin_kernel
    joint = guard.joint_asm
    assert len(res.kernels(joint)) > 40 and res.check(joint) == []      # (fp32 storage; bf16 and fp16 have code objects of their own: other_asms)
    assert guard.check(joint) == []                      # the second copy of lattice_lin_kernel (the joint translation unit's)
    assert res.check(joint.replace(".private_segment_fixed_size: 0", ".private_segment_fixed_size: 64", 1)) != []   # the check can fail
