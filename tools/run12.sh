cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
timeout 900 python tools/gemm_accuracy.py > gpurun_out/r2q/accuracy.txt 2> gpurun_out/r2q/accuracy.err; tail -3 gpurun_out/r2q/accuracy.err; cat gpurun_out/r2q/accuracy.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "full_size or stage_parity or gemm_flavours" 2>&1 | tail -4
python tools/ab.py gpurun_out/r2q 16 default
