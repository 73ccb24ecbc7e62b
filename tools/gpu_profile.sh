#!/bin/bash
# GPU-box session: kernel traces, PMC passes (separate runs: --pmc with --kernel-trace only) and bench lines per
# workload, summarised into gpurun_out/<tag>_*.  Usage: tools/gpu_profile.sh <tag> "<workloads>" [pmc]
#   e.g. tools/gpu_profile.sh r02a "c3 c4 c5 c2" pmc
set -u
TAG=${1:-r02}
WLS=${2:-"c3 c4 c5"}
PMC=${3:-}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  python "$REPO/bench.py" --workload $wl --steps 50 --warmup 5 --no-traffic-pass $( [ "$wl" = c3 ] || echo --no-cpu-baseline ) \
      > "$OUT/${TAG}_bench_${wl}.json" 2> "$OUT/${TAG}_bench_${wl}.err"
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o trace -- python "$REPO/bench.py" --workload $wl --steps 20 \
      --warmup 3 --no-cpu-baseline --no-traffic-pass > "$OUT/${TAG}_trace_${wl}.log" 2>&1
  grep "^{" "$OUT/${TAG}_trace_${wl}.log" | tail -1 > "$OUT/${TAG}_trace_${wl}.json"     # the TRACED process's own line: its HIP-event average of the dominant kernel is the one the trace must agree with
  db=$(find /tmp/prof_$wl -name "*.db" | head -1)
  [ -n "$db" ] && python "$REPO/tools/rocpd_summary.py" "$db" \
      "$TAG kernel trace: rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline" \
      > "$OUT/${TAG}_${wl}_kernel_trace.md"
  if [ -n "$PMC" ]; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_${wl}_$ctr
      rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${wl}_$ctr -o pmc -- python "$REPO/bench.py" --workload $wl \
          --steps 3 --warmup 1 --no-cpu-baseline --no-traffic-pass > "$OUT/${TAG}_pmc_${wl}_$ctr.log" 2>&1
      db=$(find /tmp/pmc_${wl}_$ctr -name "*.db" | head -1)
      if [ -n "$db" ]; then
        cp "$db" "$OUT/${TAG}_pmc_${wl}_$ctr.db"
        python "$REPO/tools/rocpd_summary.py" "$db" \
          "$TAG PMC $ctr: rocprofv3 --pmc $ctr --kernel-trace -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline" \
          > "$OUT/${TAG}_pmc_${wl}_$ctr.md"
      fi
    done
  fi
done
if [ -n "$PMC" ]; then
  specs=""
  for wl in $WLS; do
    [ -f "$OUT/${TAG}_pmc_${wl}_FETCH_SIZE.db" ] && specs="$specs $wl=$OUT/${TAG}_pmc_${wl}_FETCH_SIZE.db,$OUT/${TAG}_pmc_${wl}_WRITE_SIZE.db"
  done
  [ -n "$specs" ] && python "$REPO/tools/make_traffic.py" "$OUT/${TAG}_traffic.json" $specs > /dev/null
  rm -f "$OUT"/${TAG}_pmc_*.db
fi
ls -la "$OUT" | tail -40
