cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
( time timeout 1200 python -m pytest tests/test_gpu_mgpu.py -q -x -rs 2>&1 | tail -25 ) > gpurun_out/r2j/mgpu_tests.log 2>&1
cat gpurun_out/r2j/mgpu_tests.log
timeout 600 python bench.py --mode track --track-seconds 600 --steps 2 --warmup 1 > gpurun_out/r2j/bench_track.json 2> gpurun_out/r2j/bench_track.err
cat gpurun_out/r2j/bench_track.json; tail -3 gpurun_out/r2j/bench_track.err
