cd /tmp; export TMPDIR=/tmp
export WARP_RNNT_PATH=$GRAFT_REPO_ROOT/warp-transducer_amd/lib/dev/libwarprnnt.so
run() {  # shape dtflag tune
  rm -rf /tmp/pj; RNNT_TUNE="$3" rocprofv3 --kernel-trace --stats -d /tmp/pj -o trace -- python $GRAFT_REPO_ROOT/tools/add_network_bench.py --fused-only $2 $1 > /tmp/out.txt 2>&1
  db=$(find /tmp/pj -name "*.db" | head -1); echo "== $1 $2 $3: $(grep 'fused ' /tmp/out.txt | sed 's/.*: fused/fused/; s/(stages.*//')"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$db" "$3" | grep "rnnt::joint_d" | awk -F'|' '{print "   ", $2, $5}'
}
for shape in c2 c3 c5f32 c4 long128 long256 long1024; do for dt in "" "--bf16"; do
  run $shape "$dt" "jsplit=1"
  run $shape "$dt" "jsplit=2"
  run $shape "$dt" "jsplit=2,jfnk=2"
done; done
