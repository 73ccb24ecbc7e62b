#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], "lattice", r["stage_ms"]["lattice"], "step", r["ms_per_step"])'
for v in "" 1 2 4 8 16 3 7 31; do
  if [ -n "$v" ]; then export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/ab$v WARPRNNT_BINDING=ctypes; else unset WARP_RNNT_PATH; export WARPRNNT_BINDING=ctypes; fi
  python bench.py --workload c4 --steps 20 --no-cpu-baseline --no-traffic-pass --no-verify 2>/dev/null | python -c "$J" "ablation ${v:-none}"
  python bench.py --workload c4 --override N=1 --steps 20 --no-cpu-baseline --no-traffic-pass --no-verify 2>/dev/null | python -c "$J" "   N=1 ablation ${v:-none}"
done
