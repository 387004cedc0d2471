cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
for B in 1 4 16; do
  X=""; [ $B = 1 ] && X="--batched-lstm"
  timeout 600 python bench.py --tracks $B $X --steps 4 --warmup 2 --no-cpu-baseline --lstm-profile --serial > gpurun_out/r2b/prof_B$B.json 2> gpurun_out/r2b/prof_B$B.err
  grep "^# lstm" gpurun_out/r2b/prof_B$B.err
done
