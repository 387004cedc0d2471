#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c4
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04c4/pytest_full.log 2>&1
grep -E "passed|failed" gpurun_out/r04c4/pytest_full.log | tail -2
python bench.py --tracks 32 --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-single-track --lstm-profile > gpurun_out/r04c4/bench_prof.json 2> gpurun_out/r04c4/bench_prof.err
grep "# lstm" gpurun_out/r04c4/bench_prof.err
python tools/bench_brief.py gpurun_out/r04c4/bench_prof.json | head -12
python bench.py --tracks 48 --steps 4 --warmup 2 --no-cpu-baseline --no-pcie --no-single-track --lstm-profile > gpurun_out/r04c4/bench_prof48.json 2> gpurun_out/r04c4/bench_prof48.err
grep "# lstm alone" gpurun_out/r04c4/bench_prof48.err
python tools/bench_brief.py gpurun_out/r04c4/bench_prof48.json | head -3
python bench.py --tracks 32 --steps 8 --warmup 2 --no-cpu-baseline --no-pcie --no-single-track > gpurun_out/r04c4/bench.json 2> gpurun_out/r04c4/bench.err
python tools/bench_brief.py gpurun_out/r04c4/bench.json | head -12
