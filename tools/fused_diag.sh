#!/bin/bash
# Diagnosis of the fused gradient kernel on c4 (development build): timing of the ablations and raw FETCH_SIZE / WRITE_SIZE per launch.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/fused_diag.sh r04d'
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd "$REPO"; mkdir -p gpurun_out
export WARP_RNNT_PATH=$REPO/warp-transducer_amd/lib/dev
OUT=$REPO/gpurun_out/${TAG}_fused_diag.log
: > $OUT
for t in "fuse=0" "fuse=1,fdev=0" "fuse=1,fdev=1" "fuse=1,fdev=2" "fuse=1,fdev=3" $EXTRA_TUNES; do
  RNNT_TUNE=$t,fusemin=0 python bench.py --workload c4 --steps 30 --warmup 5 --no-cpu-baseline --no-traffic-pass --no-verify 2>/dev/null | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('c4 $t', 'ms', j['value'], 'stages', j['stage_ms'])" >> $OUT
done
cd /tmp; export TMPDIR=/tmp
for t in "fuse=0" "fuse=1,fdev=0" "fuse=1,fdev=1" $EXTRA_PMC; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcd
    RNNT_TUNE=$t,fusemin=0 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmcd -o pmc -- python $REPO/bench.py --workload c4 --steps 3 --warmup 1 \
        --no-cpu-baseline --no-traffic-pass --no-verify > /dev/null 2>&1
    db=$(find /tmp/pmcd -name "*.db" | head -1)
    [ -n "$db" ] && python - "$db" $ctr "$t" >> $OUT <<'PY'
import sys
sys.path.insert(0, sys.argv[0] and ".")
import os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from make_traffic import per_launch
per = per_launch(sys.argv[1], sys.argv[2])
print(sys.argv[3], sys.argv[2], {k: round(v / 1024 / 1024, 4) for k, v in per.items()}, "GiB raw counter per launch")
PY
  done
done
cat $OUT
