#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): the -m gpu suite, the smoke test, the default bench line and
# the profile set of tools/gpu_profile.sh.  Everything lands in gpurun_out/ (scratch); copy what should be judged to profiles/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_call.sh r02z'
set -x
TAG=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout 1500 tools/gpu_profile.sh "$TAG" "c3 c4 c5 c2" pmc > gpurun_out/profile.log 2>&1
