#!/bin/bash
# A/B on one box: the lattice kernel with and without the L2 touch loads (lib/touch = -DLAT_TOUCH=1), alternating.
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], "lattice", r["stage_ms"]["lattice"], "step", r["ms_per_step"], r["check"].get("passed"))'
export WARPRNNT_BINDING=ctypes
for rep in 1 2 3; do
  for v in base touch; do
    if [ $v = touch ]; then export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/touch; else unset WARP_RNNT_PATH; fi
    python bench.py --workload c4 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "$v c4"
    python bench.py --workload c4 --override N=16 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "$v c4-N16"
    python bench.py --workload c4 --override L=127 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "$v c4-U128"
  done
done
export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/touch
timeout 900 python -m pytest tests/test_gpu_parity_large.py tests/test_gpu_parity.py tests/test_gpu_lattice_dump.py -m gpu -q 2>&1 | tail -2
