#!/usr/bin/env python
"""Print ms/step and per-stage times of bench.py for a list of workloads (development aid).
Usage: python tools/stage_times.py c3 c5 ...   (honours RNNT_TUNE)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for w in sys.argv[1:] or ["c3"]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", w, "--no-cpu-baseline", "--no-traffic-pass",
                          "--steps", "20", "--warmup", "5", "--override", os.environ.get("OVERRIDE", "")] + (["--varlen"] if os.environ.get("VARLEN") else []), capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(w, "FAILED", out.stderr[-400:]); continue
    j = json.loads(line[-1])
    print("%s ovr=%s tune=%-24s ms=%.4f stages=%s grad=%.0fGB/s stats=%.0fGB/s path=%.3f" % (
        w, os.environ.get("OVERRIDE", "-"), os.environ.get("RNNT_TUNE", "-"), j["value"], j["stage_ms"], j["roofline"]["achieved"],
        j["stats_roofline"]["achieved"], j["path_roofline"]["frac"]))
