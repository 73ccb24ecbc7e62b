#!/bin/bash
# A/B of the XCD-contiguous chunk order in the (unfused) gradient stream, development build: RNNT_TUNE=fdev=4 against fdev=0
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/xcd_ab.sh r04h "c3 c4 c5"'
TAG=${1:-rXX}; WL=${2:-"c3 c4 c5"}
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/dev
OUT=gpurun_out/${TAG}_xcd_ab.log; : > $OUT
for w in $WL; do for rep in 1 2 3; do for f in 0 4; do
  RNNT_TUNE=fuse=0,fdev=$f python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-traffic-pass --no-verify 2>/dev/null | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$w fdev=$f', 'ms', j['value'], 'stages', j['stage_ms'])" >> $OUT
done; done; done
cat $OUT
