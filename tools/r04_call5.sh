#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c5
timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_robustness.py -x -q -m gpu > gpurun_out/r04c5/pytest_batch.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r04c5/pytest_batch.log | tail -3
python tools/ab.py gpurun_out/r04c5/ab 32,48 default env:UMX_LSTM_GROUPED=0 2>&1 | tee gpurun_out/r04c5/ab.log
AB_PROFILE=1 python tools/ab.py gpurun_out/r04c5/abp 32 default 2>&1 | tee gpurun_out/r04c5/abp.log
grep "# lstm alone" gpurun_out/r04c5/abp/default_B32.err | head -6
