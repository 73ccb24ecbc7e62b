set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export WARP_RNNT_PATH=$GRAFT_REPO_ROOT/warp-transducer_amd/lib/dev
( for t in xst=0 xst=0 ; do RNNT_TUNE=$t python tools/stage_times.py c4; done; python tools/stage_times.py c5 c2 c3 ) > gpurun_out/r02i_stage_times.log 2>&1
