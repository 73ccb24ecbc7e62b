set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/readme_table.py > gpurun_out/r02g_readme_table.md 2> gpurun_out/r02g_readme_table.err
timeout 1500 tools/gpu_profile.sh r02g "c3 c4 c5 c2" pmc > gpurun_out/profile.log 2>&1
for wl in c3 c4 c5; do python bench.py --workload $wl --varlen --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02g_bench_${wl}_varlen.json 2>/dev/null; python bench.py --workload $wl --varlen --packed --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02g_bench_${wl}_varlen_packed.json 2>/dev/null; done
