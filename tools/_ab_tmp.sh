cd /tmp; export TMPDIR=/tmp
export WARP_RNNT_PATH=$GRAFT_REPO_ROOT/warp-transducer_amd/lib/dev/libwarprnnt.so
for v in "jfsum=1" "jfsum=0" "jfsum=1,jnocb=0"; do
  rm -rf /tmp/pj; RNNT_TUNE="$v" rocprofv3 --kernel-trace --stats -d /tmp/pj -o trace -- python $GRAFT_REPO_ROOT/tools/add_network_bench.py --fused-only c4 > /dev/null 2>&1
  db=$(find /tmp/pj -name "*.db" | head -1); echo "== $v"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$db" "$v" | grep "rnnt::"
done
