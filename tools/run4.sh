cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r2f/gputests.log 2>&1
cat gpurun_out/r2f/gputests.log
python tools/ab.py gpurun_out/r2f 1,16 default
UMX_U8=dequant python tools/ab.py gpurun_out/r2f/deq 16 default
AB_BENCH_ARGS="--tracks 1" python tools/ab.py gpurun_out/r2f/single 1 default | head -3
