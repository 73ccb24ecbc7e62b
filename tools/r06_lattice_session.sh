#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|error" | tail -3
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], r["ms_per_step"], r["stage_ms"], r["check"].get("passed"))'
for i in 1 2; do python bench.py --workload c4 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4; done
python bench.py --workload c4 --varlen --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-varlen
python bench.py --workload c4 --override N=16 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-N16
python bench.py --workload c4 --override L=127 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-U128
python bench.py --workload c4 --override L=511 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-U512
python tools/add_network_bench.py --fused-only c4 long128 2>&1 | grep -v amdgpu | sed "s/| autograd.*//"
timeout 300 python tools/overlay_fuzz.py 30 61 2>&1 | tail -1
timeout 300 python tools/materialised_fuzz.py 200 9 2>&1 | tail -1 | cut -c1-200
