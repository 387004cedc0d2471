cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
( time timeout 2000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) 2>&1 | tail -8
python tools/ab.py gpurun_out/r2p 16,1 default
UMX_TAIL_STREAM=0 python tools/ab.py gpurun_out/r2p/notail 16 default
AB_BENCH_ARGS="--tracks 1" python tools/ab.py gpurun_out/r2p/single 1 default
