#!/bin/bash
# tools/configs_round.sh -- the bench line of every BASELINE configuration and batch size -> gpurun_out/r2r/ (run through gpurun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
python bench.py > gpurun_out/r2r/bench_default.json 2> gpurun_out/r2r/bench_default.err
python bench.py --tracks 4 --no-cpu-baseline --no-single-track > gpurun_out/r2r/bench_B4.json 2>/dev/null
python bench.py --tracks 16 --no-cpu-baseline --no-single-track > gpurun_out/r2r/bench_B16.json 2>/dev/null
python bench.py --tracks 48 --no-cpu-baseline --no-single-track --no-pcie > gpurun_out/r2r/bench_B48.json 2>/dev/null
python bench.py --tracks 1 --no-cpu-baseline > gpurun_out/r2r/bench_B1.json 2>/dev/null
python bench.py --tracks 32 --no-wiener --no-cpu-baseline --no-single-track --no-pcie > gpurun_out/r2r/bench_cfg2.json 2>/dev/null
python bench.py --tracks 32 --vocals-only --no-cpu-baseline --no-single-track --no-pcie > gpurun_out/r2r/bench_cfg1.json 2>/dev/null
python bench.py --tracks 1 --track-seconds 600 --no-cpu-baseline --no-pcie > gpurun_out/r2r/bench_track600.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default","bench_B48","bench_B16","bench_B4","bench_B1","bench_cfg2","bench_cfg1","bench_track600"):
    j=json.loads(open(f"gpurun_out/r2r/{f}.json").read().strip().splitlines()[0])
    l=[k for k in j["kernels"] if "lstm" in k["kernel"]][0]
    print(f, j["value"], j["ms_per_step"], "pcie", j.get("value_pcie"), "single", (j.get("single_track") or {}).get("value"), "lstm", l["launch_ms_alone"], l.get("frac_of_fp32_roof_algorithmic_alone"), l["us_per_step_alone"], "track", j.get("track"))
PY
