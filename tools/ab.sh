#!/bin/bash
# A/B harness: tools/ab.sh <tag> [lib ...]  -- pipelined and serial bench of each variant library
mkdir -p gpurun_out
for lib in "$@"; do
  tag=$(basename $lib .so)
  UMX_HIP_LIB=$lib python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  UMX_HIP_LIB=$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --serial --lstm-profile > gpurun_out/abp_$tag.json 2> gpurun_out/abp_$tag.err
  python - <<PY
import json
a=json.load(open("gpurun_out/ab_$tag.json")); b=json.load(open("gpurun_out/abp_$tag.json"))
print("$tag", "pipelined ms", a["ms_per_step"], "rec alone", a["stages_ms_unpipelined"]["lstm_rec1"], "serial+prof ms", b["ms_per_step"])
PY
  grep "lstm layer 1" gpurun_out/abp_$tag.err
done
