#!/bin/bash
# One box: the c4 placement probe plain and under four rocprofv3 --pmc passes (counters in their own runs, --kernel-trace only).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/c4_bimodal_session.sh r05b'
TAG=${1:-r05x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_bimodal
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/c4_bimodal_probe.py run 2> /dev/null > $OUT/plain.jsonl
cat $OUT/plain.jsonl
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ TCC_EA0_WRREQ"; do
  i=$((i+1))
  rm -rf /tmp/bim_$i
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace -d /tmp/bim_$i -o pmc -- python $ROOT/tools/c4_bimodal_probe.py run 2> /dev/null > $OUT/pmc_$i.jsonl )
  db=$(find /tmp/bim_$i -name "*.db" | head -1)
  [ -n "$db" ] && python tools/c4_bimodal_probe.py table $db > $OUT/pmc_$i.md
done
