set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
