cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -5
python tools/ab.py gpurun_out/r2d 1,4,16 default variants/libumx_hip_g0.so variants/libumx_hip_g4i28.so
