#!/bin/bash
# Second session: which allocation carries the state (matrix), does a tile order remove it (dev library, alternating in one process),
# and the write-side / DRAM-vs-fabric counters per placement.
TAG=${1:-r05x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_bimodal
mkdir -p $OUT
export TMPDIR=/tmp
WARP_RNNT_PATH=$ROOT/warp-transducer_amd/lib/dev RNNT_TUNE_LIVE=1 timeout 300 python tools/c4_bimodal_probe.py matrix 2> /dev/null > $OUT/matrix_orders.jsonl
cat $OUT/matrix_orders.jsonl
i=4
for set in "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
           "TCC_EA0_RDREQ TCC_EA0_WRREQ" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/bim_$i
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/bim_$i -o pmc -- python $ROOT/tools/c4_bimodal_probe.py run 2> /dev/null > $OUT/pmc_$i.jsonl )
  db=$(find /tmp/bim_$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/c4_bimodal_probe.py table $db > $OUT/pmc_$i.md
done
