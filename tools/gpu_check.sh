#!/bin/bash
# tools/gpu_check.sh [tag] -- the standard GPU cycle of a build (run through gpurun): GPU test suite, default bench line,
# accuracy table of every arithmetic flavour against float64 -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
tag=${1:-check}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1
tail -5 gpurun_out/$tag/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
timeout 600 python tools/gemm_accuracy.py > gpurun_out/$tag/accuracy.txt 2>&1
python - <<PY
import json
j=json.loads(open("gpurun_out/$tag/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms", j["ms_per_step"], "pcie", j.get("value_pcie"), "single", (j.get("single_track") or {}).get("value"))
print({k: round(v, 3) for k, v in j["stages_ms_unpipelined"].items()})
print(j.get("gemm_view"))
PY
grep -E "^fc1|^fc2|^lstm" gpurun_out/$tag/accuracy.txt | head -24
