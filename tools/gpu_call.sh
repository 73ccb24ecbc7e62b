set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
( python tools/add_network_bench.py --fused-only c2 c3 c4 c5f32; python tools/add_network_bench.py --bf16 c2 c3 c4 c5f32; python tools/add_network_bench.py --fp16 c3 c5f32 ) > gpurun_out/r02n_add_16bit.log 2>&1
cd /tmp && export TMPDIR=/tmp
for s in c3 c5f32; do rm -rf /tmp/pa_$s; rocprofv3 --kernel-trace --stats -d /tmp/pa_$s -o t -- python $GRAFT_REPO_ROOT/tools/add_network_bench.py --bf16 $s > /dev/null 2>&1; db=$(find /tmp/pa_$s -name "*.db" | head -1); [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$db" "r02n additive joint, bf16 storage, $s shape: rocprofv3 --kernel-trace --stats -- python tools/add_network_bench.py --bf16 $s" | grep -v "at::\|rocclr" > $GRAFT_REPO_ROOT/gpurun_out/r02n_add_bf16_${s}_kernel_trace.md; done
