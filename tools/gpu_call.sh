set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TA_[A-Z_0-9]+|GRBM_[A-Z_0-9]+)" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_full.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_x
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_x -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  db=$(find /tmp/pmc_x -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$db" "r02f PMC $set (c4)" > $GRAFT_REPO_ROOT/gpurun_out/r02f_pmc_c4_$tag.md
done
