#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): the -m gpu suite, the smoke test, the default bench line and
# the profile set of tools/gpu_profile.sh.  Everything lands in gpurun_out/ (scratch); copy what should be judged to profiles/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_call.sh r03s'
set -x
TAG=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout 1500 tools/gpu_profile.sh "$TAG" "c3 c4 c5 c2" pmc > gpurun_out/profile.log 2>&1
# additive joint: fp32 / bf16 / fp16 storage, kernel traces of the c3 shape; RNNTLoss through autograd
{ python tools/add_network_bench.py c3 c5f32 c4 c2; python tools/add_network_bench.py --bf16 c3 c5f32; python tools/add_network_bench.py --fp16 c3 c5f32; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_add_network_bench.log
{ python tools/add_network_bench.py --fused-only long128 long256 long1024; python tools/add_network_bench.py --bf16 c4 long128 long256 long1024; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_add_network_long.log
python tools/add_network_fuzz.py 300 11 2>&1 | tail -1 > gpurun_out/${TAG}_add_network_fuzz.log
python tools/autograd_bench.py c2 c3 c5 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_autograd_bench.log
{ python tools/binding_ab.py 1; python tools/binding_ab.py 0; } 2>&1 | grep validate > gpurun_out/${TAG}_binding_ab.log
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profj && rocprofv3 --kernel-trace --stats -d /tmp/profj -o trace -- python $OLDPWD/tools/add_network_bench.py --bf16 c3 > /dev/null 2>&1; db=$(find /tmp/profj -name "*.db" | head -1); [ -n "$db" ] && python $OLDPWD/tools/rocpd_summary.py "$db" "$TAG additive joint, bf16 storage, c3 shape: rocprofv3 --kernel-trace --stats -- python tools/add_network_bench.py --bf16 c3" > $OLDPWD/gpurun_out/${TAG}_add_bf16_c3_kernel_trace.md )
python bench.py --workload c5 --force-sharded --no-cpu-baseline --no-traffic-pass > gpurun_out/${TAG}_bench_c5_sharded_one_rank.json 2> /dev/null
python bench.py --workload c2 --pinned-costs --no-cpu-baseline --no-traffic-pass > gpurun_out/${TAG}_bench_c2_pinned.json 2> /dev/null
python bench.py --workload c4 --aux-stream --steps 50 --no-cpu-baseline --no-traffic-pass > gpurun_out/${TAG}_bench_c4_aux_stream.json 2> /dev/null
python bench.py --gpus 2 --oversubscribe-gloo --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_2ranks_oversubscribed.json 2> /dev/null
( cd /tmp && bash $OLDPWD/tools/joint_c4_profile.sh $TAG > /dev/null 2>&1 )
# round 5: a roofline per kernel of the additive joint, the first call of a process per storage type, the c4 placement matrix
( cd /tmp && for s in "c3" "--bf16 c3" "c4"; do n=$(echo $s | tr -d " -"); timeout 400 python $OLDPWD/tools/add_network_roofline.py --json $OLDPWD/gpurun_out/${TAG}_add_traffic.json $s > $OLDPWD/gpurun_out/${TAG}_add_roofline_$n.md 2>/dev/null; done )
( python tools/first_call.py x; python tools/first_call.py; python tools/first_call.py add; python tools/first_call.py add16 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_first_call.log
timeout 300 python tools/c4_bimodal_probe.py matrix 2> /dev/null > gpurun_out/${TAG}_c4_placement_matrix.jsonl
timeout 300 python tools/materialised_fuzz.py 200 7 2>&1 | tail -1 > gpurun_out/${TAG}_materialised_fuzz.log
for seed in 31 32 33; do timeout 300 python tools/add_network_fuzz.py 200 $seed 2>&1 | tail -1; done > gpurun_out/${TAG}_add_network_fuzz3.log
for seed in 11 12; do timeout 500 python tools/overlay_fuzz.py 40 $seed 2>&1 | tail -1; done > gpurun_out/${TAG}_overlay_fuzz.log
python bench.py --workload c3 --in-place --steps 30 --no-cpu-baseline --no-traffic-pass > gpurun_out/${TAG}_bench_c3_in_place.json 2> /dev/null
python tools/readme_table.py > gpurun_out/${TAG}_readme_table.md 2> gpurun_out/${TAG}_readme_table.err
