#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], "lattice", r["stage_ms"]["lattice"], "coef", r["stage_ms"]["coef"], "step", r["ms_per_step"], r["check"].get("passed"))'
for n in 8 16 32 48 64 96 128 192; do
  python bench.py --workload c4 --override N=$n --steps 20 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "c4 N=$n"
done
