#!/bin/bash
# Additive joint, c4 shape (N=64 T=1500 U=301 A=50 fp32): kernel trace + PMC passes (separate runs) of tools/add_network_bench.py --fused-only c4
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/joint_c4_profile.sh r04j'
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $REPO/tools/add_network_bench.py --fused-only c4 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_joint_c4_bench.log
rm -rf /tmp/pj; rocprofv3 --kernel-trace --stats -d /tmp/pj -o trace -- python $REPO/tools/add_network_bench.py --fused-only c4 > /dev/null 2>&1
db=$(find /tmp/pj -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/rocpd_summary.py "$db" "$TAG additive joint c4 shape, fp32: rocprofv3 --kernel-trace --stats -- python tools/add_network_bench.py --fused-only c4" > $OUT/${TAG}_joint_c4_kernel_trace.md
for ctr in FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU; do
  rm -rf /tmp/pjc; rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pjc -o pmc -- python $REPO/tools/add_network_bench.py --fused-only c4 > /dev/null 2>&1
  db=$(find /tmp/pjc -name "*.db" | head -1)
  [ -n "$db" ] && python - "$db" $ctr >> $OUT/${TAG}_joint_c4_pmc.log <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from make_traffic import per_launch
per = per_launch(sys.argv[1], sys.argv[2])
print(sys.argv[2], {k: round(v, 1) for k, v in per.items() if not k.startswith("at::") and "rocclr" not in k}, "(raw counter per launch; FETCH/WRITE in KiB)")
PY
done
cat $OUT/${TAG}_joint_c4_bench.log $OUT/${TAG}_joint_c4_kernel_trace.md $OUT/${TAG}_joint_c4_pmc.log
