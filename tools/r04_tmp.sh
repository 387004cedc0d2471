cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c8
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04c8/pytest.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r04c8/pytest.log | tail -3
python tools/ab.py gpurun_out/r04c8/ab 32,16,48 default variants/libumx_hip_nofuse.so 2>&1 | tee gpurun_out/r04c8/ab.log
