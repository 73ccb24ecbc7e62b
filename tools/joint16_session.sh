#!/bin/bash
# GPU-box session for the bf16 matrix-core joint kernels: parity tests, then A/B timings with the development build.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TAG=${1:-r03e}
( timeout 600 python -m pytest tests/test_gpu_add_network.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/${TAG}_joint16_tests.log
export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/dev
{
for tune in "j16=0" "j16=7" "j16=7,j16pf=1" ${EXTRA_TUNES}; do
  echo "== RNNT_TUNE=$tune"
  RNNT_TUNE=$tune python tools/add_network_bench.py --bf16 c3 c5f32 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/${TAG}_joint16_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof16
RNNT_TUNE=j16=7 rocprofv3 --kernel-trace --stats -d /tmp/prof16 -o trace -- python $GRAFT_REPO_ROOT/tools/add_network_bench.py --bf16 c3 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_joint16_trace.log 2>&1
db=$(find /tmp/prof16 -name "*.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$db" "$TAG additive joint bf16, c3 shape: rocprofv3 --kernel-trace --stats -- python tools/add_network_bench.py --bf16 c3 (RNNT_TUNE=j16=7)" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_add_bf16_c3_kernel_trace.md
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_joint16_tests.log $GRAFT_REPO_ROOT/gpurun_out/${TAG}_joint16_bench.log
head -16 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_add_bf16_c3_kernel_trace.md
