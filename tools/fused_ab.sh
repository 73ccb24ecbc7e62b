#!/bin/bash
# A/B of the fused gradient kernel (no coefficient kernel, no record table) on the development build:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/fused_ab.sh r04c c4'
# alternating runs of bench.py with RNNT_TUNE=fuse=0 / fuse=1 on the same box; one JSON line per run in gpurun_out/<tag>_fused_ab.log
TAG=${1:-rXX}; shift
WL=${@:-c4}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/dev
OUT=gpurun_out/${TAG}_fused_ab.log
: > $OUT
for w in $WL; do
  for rep in 1 2 3; do
    for f in 0 1; do
      RNNT_TUNE=fuse=$f,fusemin=0 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-traffic-pass 2>/dev/null | \
        python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$w fuse=$f', 'ms', j['value'], 'stages', j['stage_ms'], 'check', j['check'].get('passed'), j['check'].get('max_abs_grad_err'))" >> $OUT
    done
  done
done
cat $OUT
