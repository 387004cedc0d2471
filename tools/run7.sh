cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
( time timeout 2000 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 ) > gpurun_out/r2i/gputests.log 2>&1
cat gpurun_out/r2i/gputests.log
python tools/ab.py gpurun_out/r2i 16,1 default
UMX_GEMM=bf16x3 python tools/ab.py gpurun_out/r2i/staged 16 default
