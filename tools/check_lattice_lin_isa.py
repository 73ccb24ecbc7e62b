#!/usr/bin/env python
"""Guard for the hand-counted vmcnt of lattice_lin_kernel's operand wavefronts (csrc/rnnt_kernels.h: lattice_lin_body).

That role keeps a ring of PFD chunks x KW rows of log-probs in flight: the requests (buffer_load_dwordx2) and the waits
(s_waitcnt vmcnt((PFD-1)*KW)) are written as inline asm because the compiler's own bookkeeping across the loop's back edge
collapsed to "wait for everything".  The count is only right while NO OTHER vector-memory instruction is issued by that
role between a request and its wait: a compiler-inserted spill, reload or re-materialised load there would make the wait
return early (stale registers) without any test noticing unless its values happen to differ.  This script compiles the
device code (hipcc --cuda-device-only -S) -- or reads an existing .s -- and checks, per operand loop of the kernel:
  * the kernel uses no scratch and spills no VGPR (so the compiler has no memory traffic of its own to insert);
  * every wait inside the loop is the hand-written one, there are PFD of them per unrolled trip, and exactly KW
    hand-written requests follow each (the ring slot's refill);
  * between the loop's first request and its closing vmcnt(0) waits, EVERY vector-memory instruction is a hand-written one.
Exit code 0 = holds; 1 = violated (message says where).  `make -C warp-transducer_amd isa-check` and tests/test_isa_guard.py run it.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# both translation units instantiate the kernel (launch_lattice lives in rnnt_host.h): two code objects, two copies to check
SRCS = [os.path.join(ROOT, "warp-transducer_amd", "csrc", name) for name in ("rnnt_gpu.hip", "rnnt_joint.hip", "rnnt_gpu_f64.hip", "rnnt_gpu_h16.hip", "rnnt_joint_bf16.hip", "rnnt_joint_fp16.hip")]
NO_LIN = ("rnnt_gpu_f64.hip",)        # translation units without the linear-domain lattice kernel (it exists for fp32 lattices only)
PFD, KW = 8, 4                                   # lattice_lin_body: chunks in flight, rows per operand wavefront and chunk
VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")


def device_asm(path=None, SRC=SRCS[0]):
    if path:
        return open(path).read()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dev.s")
        cmd = [os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden",
               "-Wno-unused-lambda-capture", "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", SRC, "-o", out]
        subprocess.run(cmd, check=True)
        return open(out).read()


def check(asm):
    lines = asm.split("\n")
    problems = []
    starts = [i for i, l in enumerate(lines) if (l.startswith("_ZN4rnnt18lattice_lin_kernel") or l.startswith("_ZN4rnntL18lattice_lin_kernel")) and ":" in l]   # (L: internal linkage -- the kernel is TU-local since round 6)
    if not starts:
        return ["lattice_lin_kernel not found in the device code"]
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    meta = asm[asm.index("amdhsa.kernels"):]
    block = next((b for b in meta.split("  - .agpr_count") if "lattice_lin_kernel" in b), "")
    for key in ("private_segment_fixed_size", "vgpr_spill_count"):
        m = re.search(r"\.%s:\s+(\d+)" % key, block)
        if not m or int(m.group(1)) != 0:
            problems.append("%s = %s (must be 0: spills are vector-memory instructions the wait does not count)" % (key, m.group(1) if m else "?"))
    # events in text order: (line, kind, hand_written)
    events, in_asm = [], False
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif t and not t.startswith((";", ".")):
            op = t.split()[0]
            if VMEM.match(op):
                events.append((i, "vmem:" + op, in_asm))
            elif op == "s_waitcnt" and "vmcnt" in t and in_asm:
                events.append((i, "wait:" + re.search(r"vmcnt\((\d+)\)", t).group(1), True))
    hand_req = [e for e in events if e[2] and e[1] == "vmem:buffer_load_dwordx2"]
    ring_wait = [e for e in events if e[1] == "wait:%d" % ((PFD - 1) * KW)]
    final_wait = [e for e in events if e[1] == "wait:0"]
    if not hand_req or not ring_wait or not final_wait:
        return problems + ["hand-written requests / waits not found (requests %d, ring waits %d, final waits %d)"
                           % (len(hand_req), len(ring_wait), len(final_wait))]
    # operand loops: one per direction instantiation; split the ring waits where the gap is large
    groups, cur = [], [ring_wait[0]]
    for e in ring_wait[1:]:
        if e[0] - cur[-1][0] > 400:
            groups.append(cur); cur = []
        cur.append(e)
    groups.append(cur)
    for gi, g in enumerate(groups):
        if len(g) != PFD:
            problems.append("operand loop %d: %d ring waits per unrolled trip, expected %d" % (gi, len(g), PFD))
        closing = [e for e in final_wait if e[0] > g[-1][0]][:PFD]
        hi = closing[-1][0] if closing else g[-1][0]
        if len(closing) != PFD:
            problems.append("operand loop %d: %d closing vmcnt(0) waits, expected %d" % (gi, len(closing), PFD))
        prev_hi = groups[gi - 1][-1][0] if gi > 0 else -1
        mine = [e for e in hand_req if prev_hi < e[0] < hi]         # prologue (PFD*KW) + one refill per ring slot (PFD*KW)
        lo = mine[0][0] if mine else g[0][0]
        if len(mine) != 2 * PFD * KW:
            problems.append("operand loop %d: %d hand-written requests, expected %d (prologue) + %d (one refill per ring slot)"
                            % (gi, len(mine), PFD * KW, PFD * KW))
        foreign = [e for e in events if lo <= e[0] <= hi and e[1].startswith("vmem:") and not e[2]]
        for e in foreign:
            problems.append("operand loop %d: compiler-issued %s at line %d of the kernel between the requests and their waits"
                            % (gi, e[1][5:], e[0]))
        # exactly one refill (KW requests) between consecutive ring waits, wherever the loop rotation put the trip's seam
        for k in range(len(g) - 1):
            n = len([e for e in hand_req if g[k][0] < e[0] < g[k + 1][0]])
            if n != KW:
                problems.append("operand loop %d: %d requests between ring waits %d and %d, expected %d" % (gi, n, k, k + 1, KW))
        seam = len([e for e in mine if e[0] < g[0][0]]) - PFD * KW + len([e for e in mine if e[0] > g[-1][0]])
        if seam != KW:
            problems.append("operand loop %d: %d requests around the trip's seam, expected %d" % (gi, seam, KW))
    if len(groups) != 2:
        problems.append("%d operand loops found, expected 2 (alpha and beta instantiations)" % len(groups))
    return problems


if __name__ == "__main__":
    if len(sys.argv) > 1:
        probs = check(device_asm(sys.argv[1]))
    else:
        probs = [os.path.basename(src) + ": " + p for src in SRCS if os.path.basename(src) not in NO_LIN for p in check(device_asm(None, src))]
    if probs:
        print("lattice_lin_kernel: the hand-counted vmcnt of the operand role is NOT safe with this build:")
        for p in probs:
            print("  *", p)
        sys.exit(1)
    print("lattice_lin_kernel: operand role holds exactly its hand-written vector-memory instructions "
          "(%d requests in flight, %d per refill); no scratch, no spills (both translation units)" % (PFD * KW, KW))
