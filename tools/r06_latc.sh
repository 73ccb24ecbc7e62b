#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], "lattice", r["stage_ms"]["lattice"], "step", r["ms_per_step"], r["check"].get("passed"), r["check"].get("max_err_over_quantum"))'
for v in "" c32; do
  if [ -n "$v" ]; then export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/$v WARPRNNT_BINDING=ctypes; else unset WARP_RNNT_PATH; export WARPRNNT_BINDING=ctypes; fi
  for i in 1 2; do python bench.py --workload c4 --steps 20 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "C=${v:-16} c4"; done
  python bench.py --workload c4 --override N=1 --steps 20 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "C=${v:-16} c4 N=1"
  python bench.py --workload c4 --override L=40 --steps 20 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "C=${v:-16} c4 U=41 (one wavefront)"
  python bench.py --workload c4 --override L=40,N=300 --steps 20 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "C=${v:-16} T=1500 U=41 N=300 (log-domain one-wavefront kernel)"
  python tools/add_network_bench.py --fused-only c4 2>&1 | grep -v amdgpu | sed "s/| autograd.*//"
done
