#!/usr/bin/env python3
"""ISA guard for the hand-issued vector-memory accesses of the lattice kernels (csrc/rnnt_kernels.h: LatIO, lattice_body).

The compiler does not look into an asm block, so two things it normally guarantees are checked here on the generated code of
every translation unit that holds such accesses (hipcc --cuda-device-only -S, or existing .s files given on the command line):

  1. SGPR hazard.  A vector-memory instruction must not read an SGPR (descriptor, scalar offset) that a VECTOR instruction
     (v_readfirstlane, v_readlane, a compare into an SGPR pair) wrote less than five wait states earlier.
  2. Rows in flight.  Between a hand-issued load and the `s_waitcnt vmcnt(N)` that covers it (N counted over the hand-issued
     loads and stores that follow it), no instruction may read or write the load's destination registers -- a copy the
     register allocator slipped in would move a value that has not arrived.

Both scans walk each kernel linearly (the kernels' hot loops are laid out in program order; blocks placed out of line are
prologue code with no row in flight).  Exit status 0 and one summary line when clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "warp-transducer_amd", "csrc")
UNITS = ["rnnt_gpu.hip", "rnnt_gpu_h16.hip", "rnnt_gpu_f64.hip", "rnnt_joint.hip", "rnnt_joint_bf16.hip", "rnnt_joint_fp16.hip"]
NEED = 5                                   # wait states between a VALU write of an SGPR and a VMEM read of it


def device_asm(unit, tmp):
    out = os.path.join(tmp, unit + ".s")
    cmd = [os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden",
           "--cuda-device-only", "-S", os.path.join(CSRC, unit), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def sregs(text):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    if re.search(r"\bvcc(_lo)?\b", text):          # the register allocator hands vcc out as a scratch scalar too
        out.add(106)
    if re.search(r"\bvcc(_hi)?\b", text):
        out.add(107)
    return out


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(text):
    """(name, [instruction or marker lines]) per kernel of the device assembly `text` that holds hand-issued vector-memory instructions."""
    name, body, out = None, [], []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"):
            if any(x == "#ASM" for x in body) and any(x.startswith("buffer_") for x in body):
                out.append((name, body))
            name = None
            continue
        if t.startswith(";;#ASMSTART"):
            body.append("#ASM")
        elif t.startswith(";;#ASMEND"):
            body.append("#END")
        elif line.startswith("\t") and t and not t.startswith(".") and not t.startswith(";"):
            body.append(t.split(";")[0].strip())
    return out


def check_kernel(name, body):
    problems, hand, n_hand = [], False, 0
    inflight = []                      # hand-issued accesses since the oldest uncovered load: ("L", regs) / ("S", set())
    for i, ins in enumerate(body):
        if ins == "#ASM":
            hand = True
            continue
        if ins == "#END":
            hand = False
            continue
        op = ins.split()[0]
        is_vmem = op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_")
        if hand and is_vmem:
            n_hand += 1
            # 1. SGPR hazard
            reads, states, k = sregs(ins), 0, i - 1
            while k >= 0 and states < NEED:
                prev = body[k]
                k -= 1
                if prev in ("#ASM", "#END"):
                    continue
                pop = prev.split()[0]
                if pop == "s_nop":
                    states += int(prev.split()[1], 0) + 1
                    continue
                if pop.startswith("v_"):
                    dst = prev[len(pop):].split(",")[0]
                    if sregs(dst) & reads:
                        problems.append("%s: `%s` reads an SGPR that `%s` wrote %d wait state(s) earlier" % (name, ins, prev, states))
                states += 1
        # 2. rows in flight
        pending = set().union(*[r for kind, r in inflight if kind == "L"]) if inflight else set()
        if hand and is_vmem:
            if op.startswith("buffer_load"):
                first = ins[len(op):].split(",")[0]
                if vregs(ins[len(op) + len(first):]) & pending:
                    problems.append("%s: `%s` takes its address from a row in flight" % (name, ins))
                inflight.append(("L", vregs(first)))
            else:
                if vregs(ins) & pending:
                    problems.append("%s: `%s` stores a row in flight" % (name, ins))
                if inflight:
                    inflight.append(("S", set()))
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
        if m:
            n = int(m.group(1))
            inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
            if n == 0:
                inflight = []
            while inflight and inflight[0][0] == "S":
                inflight.pop(0)
            continue
        if is_vmem:                         # a compiler-issued access only makes the counted waits stricter
            if inflight:
                inflight.append(("S", set()))
            continue
        if op.startswith("s_"):
            continue
        hit = vregs(ins) & pending
        if hit and op == "v_readfirstlane_b32":
            # how the compiler materialises an UNDEFINED scalar (a loop-carried row offset on the path that leaves the loop): any
            # vector register will do, the result is dead.  A misplaced copy of a row would be a v_mov / v_accvgpr_write.
            continue
        if hit:
            problems.append("%s: `%s` touches v%s while its load is in flight" % (name, ins, sorted(hit)))
    return n_hand, problems


def check(text):
    """(lattice kernels seen, hand-issued accesses, problems) of one translation unit's device assembly."""
    problems, total, seen = [], 0, 0
    for name, body in kernels(text):
        if "lattice" not in name:
            continue
        n, p = check_kernel(name, body)
        if "lattice_lin_kernel" in name:      # its operand role has its own guard (check_lattice_lin_isa.py); here: the log-domain fallback inside it
            p = [x for x in p if "in flight" not in x]
        seen += 1
        total += n
        problems += p
    return seen, total, problems


def main(argv):
    problems, total, seen = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in argv or [device_asm(u, tmp) for u in UNITS]:
            k, n, p = check(open(f).read())
            seen, total, problems = seen + k, total + n, problems + p
    for p in problems:
        print("HAZARD " + p)
    if problems or not total:
        print("lattice asm hazards: %d problem(s) in %d kernel(s), %d hand-issued accesses" % (len(problems), seen, total))
        return 1
    print("lattice asm hazards: none in %d kernel(s), %d hand-issued accesses (SGPR wait states, rows in flight)" % (seen, total))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
