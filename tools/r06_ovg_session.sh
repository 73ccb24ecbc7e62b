#!/bin/bash
# dev build: the overlay guard of the tiled coefficient kernel, mode by mode (RNNT_TUNE=ovg=0|1|2), c4
cd "${GRAFT_REPO_ROOT:-.}"
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], r["ms_per_step"], r["stage_ms"], r["check"].get("passed"))'
export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/dev WARPRNNT_BINDING=ctypes
for i in 1 2; do for m in 0 1 2; do RNNT_TUNE=ovg=$m python bench.py --workload c4 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" "c4 ovg=$m"; done; done | tee gpurun_out/r06h_ovg.log
