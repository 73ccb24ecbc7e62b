set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_sharded_rccl.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log
