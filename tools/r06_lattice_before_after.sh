#!/bin/bash
# A/B on one box, alternating: the library at the round's coefficient-kernel commit (lib/old, built from `git archive eb51b2e`)
# against the shipped one -- lattice stage of long utterances, fp32 and fp64 lattices.
cd "${GRAFT_REPO_ROOT:-.}"
export WARPRNNT_BINDING=ctypes
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/old; else unset WARP_RNNT_PATH; fi
    for o in "fp32 64,1500,301,50" "fp32 16,1500,301,50" "fp32 64,1500,128,50" "fp32 64,1500,512,50" "fp32 24,1500,1024,50" "fp64 16,1500,301,50" "fp64 32,1500,128,50" "fp64 16,1500,512,50" "fp64 8,1500,1024,50" "fp64 32,1500,64,50"; do
      echo "$v $(python tools/lattice_stage_time.py $o 2>&1 | tail -1)"
    done
  done
done
