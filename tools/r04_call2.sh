#!/bin/bash
# round 4, GPU call 2: staging dealt to both wave groups (A/B against the round-3 form), store-once overlap-add, probe modes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c2
tools/lds_probe > gpurun_out/r04c2/lds_probe.log 2>&1
cat gpurun_out/r04c2/lds_probe.log
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -x -q -m gpu -k "ping_pong or fused or ragged or straddle or lane_companions or batch_of_tracks or whole_tracks or roundtrip" > gpurun_out/r04c2/pytest_subset.log 2>&1
tail -4 gpurun_out/r04c2/pytest_subset.log
python tools/ab.py gpurun_out/r04c2/ab 32 default variants/libumx_hip_dma0.so 2>&1 | tee gpurun_out/r04c2/ab.log
bash tools/profile_round.sh r04_v1 2>&1 | tail -3
