set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -c "gfx950" > gpurun_out/ngpu.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/lattice_bench.hip -o /tmp/lattice_bench 2>&1 | tail -3
( for u in 41 64 65 128 129 192 256 301 320 512 600 1024; do /tmp/lattice_bench 64 1500 $u; done ) > gpurun_out/r02a_lattice_bench.log 2>&1
timeout 1500 tools/gpu_profile.sh r02a "c3 c4 c5 c2" pmc > gpurun_out/profile.log 2>&1
