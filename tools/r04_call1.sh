#!/bin/bash
# round 4, GPU call 1: what limits the ping-pong GEMM (VERDICT r3 item 1 a-c) + the round's starting profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c1
tools/lds_probe > gpurun_out/r04c1/lds_probe.log 2>&1
cat gpurun_out/r04c1/lds_probe.log
python tools/ab.py gpurun_out/r04c1/ab 32 default variants/libumx_hip_ppnodma.so variants/libumx_hip_ppnomfma.so variants/libumx_hip_ppnofrag.so variants/libumx_hip_ppnoepi.so variants/libumx_hip_ppnodmanoepi.so 2>&1 | tee gpurun_out/r04c1/ab.log
bash tools/pmc_gemm.sh --tracks 32 2>&1 | tee gpurun_out/r04c1/pmc_gemm.log
bash tools/profile_round.sh r04_v0 2>&1 | tail -3
