set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -30 > gpurun_out/r2a/batch_tests.log
cat gpurun_out/r2a/batch_tests.log
for B in 1 4 16; do
  timeout 600 python bench.py --tracks $B --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_B$B.json 2> gpurun_out/r2a/bench_B$B.err
  tail -c 3000 gpurun_out/r2a/bench_B$B.json; tail -5 gpurun_out/r2a/bench_B$B.err
done
timeout 600 python bench.py --tracks 1 --batched-lstm --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2a/bench_B1b.json 2> gpurun_out/r2a/bench_B1b.err
tail -c 1500 gpurun_out/r2a/bench_B1b.json
