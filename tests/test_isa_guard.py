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


def test_hand_issued_lattice_accesses_have_no_sgpr_hazard_and_no_row_touched_in_flight(device_asm):
    """The fp32 lattice kernels issue their row loads and stores as inline assembly with exact `s_waitcnt vmcnt` (LatIO): the
    compiler's hazard recogniser and wait insertion do not look inside.  tools/check_lattice_asm_hazards.py checks the ISA of every
    translation unit for the two things that can go wrong -- and must trip on the form that did (a v_readfirstlane of a row offset
    right in front of the access: 139 GPU tests failed on it in round 6) and on a copy of a row that has not arrived."""
    guard, asm = device_asm
    import check_lattice_asm_hazards as haz
    units = {"rnnt_gpu": asm, "rnnt_joint": guard.joint_asm}
    units.update(guard.other_asms)
    accesses = 0
    for name, text in units.items():
        seen, n, problems = haz.check(text)
        assert problems == [], (name, problems[:3])
        accesses += n
    assert accesses > 1000
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4rnntL14lattice_kernelIfLi4ELi2EE") and ":" in l)
    acc = [i for i in range(start, len(lines)) if lines[i].lstrip().startswith("buffer_load_dwordx4") and ";;#ASMSTART" in "".join(lines[i - 3:i])]
    i = acc[20]
    soff = lines[i].split(",")[-1].split()[0]                      # its scalar offset register
    bad = list(lines)
    bad.insert(i, "\tv_readfirstlane_b32 %s, v5" % soff)           # directly in front of the access, behind the block's own s_mov
    assert any("wait state" in p for p in haz.check("\n".join(bad))[2])
    dst = lines[i].split()[1].rstrip(",")                          # v[a:b]
    lo = int(dst[2:].split(":")[0])
    bad = list(lines)
    bad.insert(i + 2, "\tv_mov_b32_e32 v250, v%d" % lo)            # a copy right after the request
    assert any("in flight" in p for p in haz.check("\n".join(bad))[2])
