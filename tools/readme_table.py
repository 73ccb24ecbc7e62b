#!/usr/bin/env python
"""Reproduce the reference README's "GPU Performance" table (README.md:11-32, GTX 1080 Ti numbers) on an MI355X.

For every cell N in {1,16,32,64,128} x {T150/L40/A28, T150/L20/A5000, T1500/L300/A50}:
  * reference protocol: warp-transducer_amd/build/test_time B T L A (= tests/test_time.cu:89-128: no warm-up, mean of
    10 wall-clock compute_rnnt_loss calls incl. the costs D2H copy and the stream sync);
  * steady state: bench.py --override N=..  (5 warm-up + 50 timed steps): median ms/batch and the fraction of the
    HBM roofline of the whole path, (3*E*s + 48*R) / t / 8 TB/s.
Prints a markdown table (stdout).  Usage: python tools/readme_table.py > profiles/rNN_readme_table.md"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("T=150, L=40, A=28", "c2", 150, 40, 28, {1: 8.51, 16: 11.43, 32: 12.65, 64: 14.75, 128: 19.48}),
          ("T=150, L=20, A=5000", "c3", 150, 20, 5000, {1: 4.79, 16: 24.44, 32: 41.38, 64: 80.44, 128: 51.46}),
          ("T=1500, L=300, A=50 (commented out in the README)", "c4", 1500, 300, 50,
           {1: 570.33, 16: 768.57, 32: 955.05, 64: 569.34, 128: None})]


def main():
    exe = os.path.join(ROOT, "warp-transducer_amd", "build", "test_time")
    print("# README table of the reference (README.md:11-32) on one MI355X\n")
    print("`test_time` = the reference's protocol (tests/test_time.cu:89-128: no warm-up, mean of 10 wall-clock calls, "
          "variance in brackets); `steady` = bench.py, median of 50 warmed steps (steps under 0.5 ms: of the loop without the per-stage HIP events); `roofline` = (3·E·s + 48·R) / steady / "
          "8 TB/s; `published` = GTX 1080 Ti, README.md:11-32 (N=128, A=5000 is physically an OOM run: BASELINE.md §1).\n")
    for title, wl, T, L, A, pub in SHAPES:
        print("| **%s** | published ms | test_time mean ms | steady median ms (p10–p90) | roofline | speed-up vs published |" % title)
        print("|---|---|---|---|---|---|")
        for N in (1, 16, 32, 64, 128):
            tt = subprocess.run([exe, str(N), str(T), str(L), str(A)], capture_output=True, text=True)
            m = re.search(r"average 10 time cost: ([0-9.]+) ms variance: ([0-9.]+)", tt.stdout)
            tt_s = "%.3f (%.4f)" % (float(m.group(1)), float(m.group(2))) if m else "failed"
            b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--override", "N=%d" % N,
                                "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-traffic-pass"], capture_output=True, text=True)
            line = [l for l in b.stdout.splitlines() if l.startswith("{")]
            if line:
                j = json.loads(line[-1])
                st = j.get("plain_step_ms") or j["step_ms"]        # small problems: the loop without the per-stage events
                med = st["median"]
                steady = "%.4f (%.4f–%.4f)" % (med, st["p10"], st["p90"])
                frac = "%.3f" % (j["path_roofline"]["bytes_algo"] / (med * 1e-3) / 8e12)
                sp = "%.0f×" % (pub[N] / med) if pub[N] else "—"
            else:
                steady, frac, sp = "failed", "—", "—"
            print("| N=%d | %s | %s | %s | %s | %s |" % (N, pub[N] if pub[N] else "—", tt_s, steady, frac, sp))
        print()


if __name__ == "__main__":
    main()
