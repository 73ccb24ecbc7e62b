#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], r["ms_per_step"], r["stage_ms"]["coef"], r["check"].get("passed"))'
for rep in 1 2; do
for v in "" w5k4 w4k8 w5k8 w3k8 w6k4; do
  if [ -n "$v" ]; then export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/$v WARPRNNT_BINDING=ctypes; else unset WARP_RNNT_PATH; export WARPRNNT_BINDING=ctypes; fi
  python bench.py --workload c4 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "c4 ${v:-w4k4}"
  python tools/add_network_bench.py --fused-only c4 2>&1 | grep -v amdgpu | sed "s/| autograd.*//;s/^/   ${v:-w4k4} /"
done; done
