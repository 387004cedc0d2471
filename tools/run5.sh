cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
( time timeout 2000 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r2g/gputests.log 2>&1
cat gpurun_out/r2g/gputests.log
