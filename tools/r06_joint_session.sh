#!/bin/bash
# Round 6 GPU session for the additive joint: FETCH_SIZE calibration on known byte counts, A/B of the XCD-aware block order
# (lib/base = the library before it) and the per-kernel rooflines.   gpurun --timeout 1500 -- 'bash tools/r06_joint_session.sh r06c'
TAG=${1:-r06c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. calibration
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$ctr
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/cal_$ctr -o cal -- $REPO/tools/microbench/fetch_calib > $OUT/${TAG}_calib_$ctr.log 2>&1
  db=$(find /tmp/cal_$ctr -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/rocpd_summary.py "$db" "$TAG calibration, rocprofv3 --pmc $ctr --kernel-trace -- tools/microbench/fetch_calib (every kernel moves 1073.74 MB once)" > $OUT/${TAG}_calib_$ctr.md
done
# 2. A/B: base library vs this tree, alternating
for i in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export WARP_RNNT_PATH=$REPO/warp-transducer_amd/lib/base WARPRNNT_BINDING=ctypes; else unset WARP_RNNT_PATH; export WARPRNNT_BINDING=ctypes; fi
    echo "== $which run $i" >> $OUT/${TAG}_ab.log
    python $REPO/tools/add_network_bench.py --fused-only c3 c5f32 c4 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_ab.log
    python $REPO/tools/add_network_bench.py --bf16 c3 c5f32 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_ab.log
  done
done
unset WARP_RNNT_PATH WARPRNNT_BINDING
# 3. correctness of the new order
( cd $REPO && timeout 900 python -m pytest tests/test_gpu_add_network.py -m gpu -q -x 2>&1 | tail -5 ) > $OUT/${TAG}_pytest_add.log
# 4. rooflines per kernel, new and base
for s in "c3" "--bf16 c3"; do n=$(echo $s | tr -d " -"); timeout 400 python $REPO/tools/add_network_roofline.py $s > $OUT/${TAG}_add_roofline_$n.md 2>/dev/null; done
export WARP_RNNT_PATH=$REPO/warp-transducer_amd/lib/base WARPRNNT_BINDING=ctypes
timeout 400 python $REPO/tools/add_network_roofline.py c3 > $OUT/${TAG}_add_roofline_c3_base.md 2>/dev/null
ls -la $OUT | tail -20
