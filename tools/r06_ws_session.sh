#!/bin/bash
# Round 6: the -m gpu suite and the long-utterance workloads on the per-sample-block workspace layout.
TAG=${1:-r06f}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/${TAG}_pytest_gpu.log
tail -12 gpurun_out/${TAG}_pytest_gpu.log
J='import json,sys; r=json.loads(sys.stdin.read()); print(sys.argv[1], r["ms_per_step"], r["stage_ms"], r["check"].get("passed"))'
{
for i in 1 2; do python bench.py --workload c4 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4; done
python bench.py --workload c4 --aux-stream --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-aux
python bench.py --workload c4 --varlen --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-varlen
python bench.py --workload c4 --varlen --packed --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c4-varlen-packed
python bench.py --workload c5_full --steps 10 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c5_full
python bench.py --workload c3 --steps 30 --no-cpu-baseline --no-traffic-pass 2>/dev/null | python -c "$J" c3
python tools/add_network_bench.py --fused-only c4 c3 2>&1 | grep -v amdgpu
} | tee gpurun_out/${TAG}_bench.log
