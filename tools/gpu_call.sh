set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log
( for f in "" "--pinned-costs"; do for w in c2 c5 c3; do python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline $f | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:60], '| value', d['value'], '| median', d['step_ms']['median'], '| plain', d['plain_step_ms'])"; done; done ) > gpurun_out/r02j_pinned.log 2>&1
