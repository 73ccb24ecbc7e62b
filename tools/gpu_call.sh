set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( python tools/stage_times.py c4; python tools/stage_times.py c4 ) > gpurun_out/r02p_coef_occ.log 2>&1
