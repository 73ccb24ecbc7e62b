#!/bin/bash
# dev build: columns per lane / operand ping-pong of the additive joint's DF and DG kernels on the c3 shape (fp32), RNNT_TUNE sweep
cd "${GRAFT_REPO_ROOT:-.}"
export WARP_RNNT_PATH=$PWD/warp-transducer_amd/lib/dev WARPRNNT_BINDING=ctypes
for t in "" "jgnk=2" "jgnk=1" "jgpf=0" "jgnk=2,jgpf=0" "jfnk=2" "jfnk=2,jfpf=0" "jfpf=0" "jzs=4" "jsamp=0"; do
  echo "== RNNT_TUNE=$t"
  RNNT_TUNE=$t python tools/add_network_bench.py --fused-only c3 c5f32 2>&1 | grep -v amdgpu | sed 's/| autograd.*//'
done | tee gpurun_out/r06k_joint_tune.log
