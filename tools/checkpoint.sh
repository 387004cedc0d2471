#!/bin/bash
# the evidence of a build (rounds 4-5: tools/r04_checkpoint.sh <tag>) -- GPU suite, bench line at the default 64 lanes, rocprofv3 kernel
# stats + FETCH/WRITE of the same command, SQ counters of the GEMM / recurrence kernels at 32 lanes, accuracy table; ~19 GPU-minutes
tag=${1:-r04_v2}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/$tag/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/$tag/pytest_gpu.log | tail -2
bash tools/profile_round.sh $tag 2>&1 | tail -2
bash tools/pmc_gemm.sh --tracks 32 > gpurun_out/$tag/pmc_sq_counters.txt 2>&1
timeout 900 python tools/gemm_accuracy.py > gpurun_out/$tag/accuracy_vs_float64.txt 2>&1
tail -30 gpurun_out/$tag/accuracy_vs_float64.txt
python tools/bench_brief.py gpurun_out/prof_$tag/bench.json | head -12
