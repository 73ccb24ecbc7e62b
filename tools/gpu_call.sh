set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_reference_sources.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
( oracle/_ref/ref_test_gpu; oracle/_ref/ref_test_time 16 150 40 28 ) > gpurun_out/r02k_reference_gpu_programs.log 2>&1
