#!/bin/bash
# A/B on one box: lattice row traffic as one burst per chunk (lib/spread0) or one pair per step (lib/spread1), by lattice width.
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], "lattice", r["stage_ms"]["lattice"], "step", r["ms_per_step"], r["check"].get("passed"))'
export WARPRNNT_BINDING=ctypes
for rep in 1 2; do
  for v in 0 1; do
    export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/spread$v
    for o in "N=64" "N=16" "L=127" "L=199" "L=511" "L=699,N=32" "L=1023,N=24"; do
      python bench.py --workload c4 --override $o --steps 20 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "spread=$v c4 $o"
    done
  done
done
