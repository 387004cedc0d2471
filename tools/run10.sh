cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_batch.py -q -x 2>&1 | tail -4
python tools/ab.py gpurun_out/r2l 16,4 default
