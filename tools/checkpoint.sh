#!/bin/bash
# tools/checkpoint.sh <tag>: the evidence of a build, the LAST GPU action of a round -- GPU suite, the default bench line (64 lanes, with the
# track / reset-mode legs and the CPU baseline), rocprofv3 kernel stats + FETCH / WRITE of the same command, SQ counters of the GEMM /
# recurrence / split kernels at the bench's lane count, accuracy table against float64; ~20 GPU-minutes.  Copy what is to be judged from
# gpurun_out/<tag>/ and gpurun_out/prof_<tag>/ into profiles/.
tag=${1:-r06_v1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
echo ${HEAD_SHA:-unknown} > gpurun_out/$tag/head.txt # (the snapshot on the GPU box has no .git: HEAD_SHA=$(git rev-parse HEAD) gpurun -- 'HEAD_SHA=... bash tools/checkpoint.sh')
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/$tag/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/$tag/pytest_gpu.log | tail -2
bash tools/profile_round.sh $tag 2>&1 | tail -2
bash tools/pmc_gemm.sh --tracks 64 > gpurun_out/$tag/pmc_sq_counters.txt 2>&1
timeout 900 python tools/gemm_accuracy.py > gpurun_out/$tag/accuracy_vs_float64.txt 2>&1
# in-kernel tile profile of the persistent GEMM (timing build variants/libumx_hip_psprof.so = -DPS_PROFILE=1, if it was built)
if [ -f variants/libumx_hip_psprof.so ]; then UMX_HIP_LIB=$PWD/variants/libumx_hip_psprof.so python tools/which_gemm.py 64 2>&1 | grep -E "# ps" | sort > gpurun_out/$tag/ps_tile_profile_raw.txt; fi
tail -30 gpurun_out/$tag/accuracy_vs_float64.txt
python tools/bench_brief.py gpurun_out/prof_$tag/bench.json | head -12
