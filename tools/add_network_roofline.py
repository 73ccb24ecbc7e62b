#!/usr/bin/env python
"""A roofline per KERNEL of the additive joint: for joint_z* / joint_df* / joint_dg* (and the lattice /
coefficient kernels beside them) on one shape,
  * matrix-core rate: analytic contraction flops (2*N*T*U*A per kernel) / kernel time against the dense MFMA peak of the
    operand type (fp32 157.3 TF, bf16 2500 TF: MI355X_MICROARCH.md), cross-checked by the hardware's own count
    (SQ_INSTS_VALU_MFMA_MOPS_* x 512 flops);
  * HBM rate: (2 x FETCH_SIZE + WRITE_SIZE) KiB / kernel time against 8 TB/s;
  * issue picture: vector instructions and MFMA instructions per wavefront, VALU-issue cycles (4 per instruction) and MFMA-busy
    cycles (SQ_VALU_MFMA_BUSY_CYCLES) as a fraction of the SIMD-cycles the kernel had (duration x clock x SIMDs).
Every counter set is its own `rocprofv3 --pmc ... --kernel-trace` run of tools/add_network_bench.py --fused-only (no other
trace domain), as the guide prescribes.
`--json FILE` also writes the counted HBM bytes per kernel launch and per step (launches of a kernel / launches of joint_prep_kernel or
of the lattice kernel = launches per step) -- what bench.py reports as `other_workloads.add_*.roofline.traffic`.
Usage: python tools/add_network_roofline.py [--bf16] [--json FILE] c3|c4|c5f32 > profiles/rNN_add_roofline_<shape>.md"""
import glob
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"c2": (16, 150, 41, 28), "c3": (128, 150, 21, 5000), "c4": (64, 1500, 301, 50), "c5f32": (128, 200, 41, 1024)}
CLOCK_HZ, SIMDS = 2.4e9, 1024
PASSES = [[], ["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAVES", "SQ_BUSY_CYCLES"],
          ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
          ["SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16"]]


def short(name):
    return re.sub(r"<.*", "", re.sub(r"\(.*", "", name).replace("void ", "")).replace("rnnt::", "")


def one_pass(counters, argv, tmp):
    out = os.path.join(tmp, "p_" + ("_".join(counters) or "trace")[:40])
    cmd = ["rocprofv3"] + (["--pmc"] + counters if counters else ["--stats"]) + ["--kernel-trace", "-d", out, "-o", "run", "--", sys.executable,
                                                                                 os.path.join(ROOT, "tools", "add_network_bench.py"), "--fused-only"] + argv
    r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), capture_output=True, text=True, timeout=900)
    dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        return None
    db = sqlite3.connect(dbs[0])
    res = {}
    if not counters:
        for name, n, avg in db.execute("select name, count(*), avg(end-start) from kernels group by name"):
            k = res.setdefault(short(name), {"calls": 0, "ns": 0.0})
            k["ns"] = (k["ns"] * k["calls"] + avg * n) / (k["calls"] + n)
            k["calls"] += n
    else:
        for name, cname, n, val in db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by 1, 2"):
            k = res.setdefault(short(name), {})
            acc = k.setdefault(cname, [0, 0.0])
            acc[0] += n
            acc[1] += val
        res = {k: {c: v[1] / max(v[0], 1) for c, v in d.items()} for k, d in res.items()}
    return res


def main():
    argv = sys.argv[1:]
    json_path = None
    if "--json" in argv:
        i = argv.index("--json")
        json_path = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")] or ["c3"]
    half = [a for a in argv if a in ("--bf16", "--fp16")]
    shape = args[0]
    N, T, U, A = SHAPES[shape]
    peak_tf = 2500.0 if half else 157.3
    tmp = tempfile.mkdtemp(prefix="addroof_", dir="/tmp")
    try:
        data = [one_pass(c, half + [shape], tmp) for c in PASSES]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    trace = data[0] or {}
    merged = {}
    for d in data[1:]:
        for k, v in (d or {}).items():
            merged.setdefault(k, {}).update(v)
    flops = 2.0 * N * T * U * A
    print("# additive joint, %s shape N=%d T=%d U=%d A=%d, %s storage: one roofline per kernel\n" % (shape, N, T, U, A, (half or ["fp32"])[0].strip("-")))
    print("analytic contraction flops per GEMM kernel 2*N*T*U*A = %.3g; matrix peak %.1f TF (dense, MI355X_MICROARCH.md), HBM 8000 GB/s; "
          "SIMD-cycles = duration x %.1f GHz x %d SIMDs; VALU issue = 4 cycles per vector instruction of a wavefront\n" % (flops, peak_tf, CLOCK_HZ / 1e9, SIMDS))
    print("| kernel | calls | avg us | MFMA TF/s (analytic) | of peak | MFMA TF/s (MOPS x 512) | HBM GB/s | of 8 TB/s | VALU / wave | MFMA / wave | "
          "VALU per MFMA | VALU-issue share | MFMA-busy share |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, t in sorted(trace.items(), key=lambda kv: -kv[1]["ns"] * kv[1]["calls"]):
        if k.startswith("at::") or "rocclr" in k.lower() or "elementwise" in k or "Memset" in k or "fill" in k.lower():
            continue
        m = merged.get(k, {})
        sec = t["ns"] * 1e-9
        gemm = any(s in k for s in ("joint_z", "joint_df", "joint_dg"))
        tf = flops / sec / 1e12 if gemm else float("nan")
        mops = sum(m.get(c, 0.0) for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16"))
        hbm = (2 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024
        waves = m.get("SQ_WAVES", 0.0)
        valu, mfma = m.get("SQ_INSTS_VALU", 0.0), m.get("SQ_INSTS_MFMA", 0.0)
        simd_cycles = sec * CLOCK_HZ * SIMDS
        print("| `%s` | %d | %.1f | %s | %s | %.1f | %.0f | %.3f | %.0f | %.0f | %s | %.2f | %.2f |"
              % (k[:60], t["calls"], t["ns"] / 1e3, "%.1f" % tf if gemm else "-", "%.3f" % (tf / peak_tf) if gemm else "-",
                 mops * 512 / sec / 1e12, hbm / sec / 1e9, hbm / sec / 1e9 / 8000.0, valu / max(waves, 1), mfma / max(waves, 1),
                 "%.0f" % (valu / mfma) if mfma else "-", 4.0 * valu / simd_cycles, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles))
    if json_path:
        import json
        steps = max([t["calls"] for k, t in trace.items() if "lattice" in k] or [1])
        per = {}
        for k, t in trace.items():
            m = merged.get(k, {})
            if "FETCH_SIZE" not in m and "WRITE_SIZE" not in m:
                continue
            if k.startswith("at::") or "rocclr" in k.lower() or "elementwise" in k:
                continue
            b = (2 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024
            per[k] = {"launches_per_step": t["calls"] / steps, "traffic_bytes_per_launch": int(b), "avg_us": round(t["ns"] / 1e3, 2),
                      "traffic_bytes_per_step": int(b * t["calls"] / steps)}
        key = "add_%s_%s" % (shape, (half or ["f32"])[0].strip("-").replace("fp32", "f32"))
        try:
            doc = json.load(open(json_path))
        except (OSError, ValueError):
            doc = {}
        doc[key] = {"shape": [N, T, U, A], "kernels": per, "traffic_bytes_per_step": int(sum(v["traffic_bytes_per_step"] for v in per.values())),
                    "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace, separate passes; 2 x FETCH_SIZE + WRITE_SIZE KiB "
                           "(the x2 holds for every read shape of these kernels: profiles/r06/fetch_calibration.md)"}
        json.dump(doc, open(json_path, "w"), indent=1, sort_keys=True)
    print("\n(counters are per launch, averaged over the launches of the run; `VALU-issue share` counts 4 SIMD-cycles per vector instruction "
          "-- 1.0 = the vector pipes never idle; `MFMA-busy share` = SQ_VALU_MFMA_BUSY_CYCLES over the same SIMD-cycles)")


if __name__ == "__main__":
    main()
