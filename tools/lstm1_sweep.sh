#!/bin/bash
# tools/lstm1_sweep.sh <outdir> <env-setting> ...   (an env-setting is "NAME=VALUE[,NAME=VALUE]" or "default")
# The one-track engine (bench.py --tracks 1) per setting: segments back to back, one segment alone, the recurrence's launch
# alone, and the in-kernel phase profile of the recurrence (a second run: the profiler slows the profiled chain).
out=$1; shift
mkdir -p $out
for v in "$@"; do
  tag=$(echo $v | tr -c 'A-Za-z0-9_\n' '_')
  envs=""; [ "$v" != default ] && envs=$(echo $v | tr ',' ' ')
  env $envs python bench.py --tracks 1 --steps 12 --warmup 3 --no-cpu-baseline --no-pcie > $out/$tag.json 2> $out/$tag.err
  env $envs python bench.py --tracks 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pcie --lstm-profile > $out/${tag}_prof.json 2> $out/${tag}_prof.err
  python - <<PY
import json
a = json.load(open("$out/$tag.json"))
al = a["stages_ms_unpipelined"]
rec = sum(al[f"lstm_rec{l}"] for l in range(3)) / 3
print(f"$v: back-to-back {a['ms_per_step']:.3f} ms/seg  lone {a['ms_per_step_unpipelined']:.3f} ms  recurrence alone {rec:.4f} ms/launch = {rec * 1e3 / a['config']['frames']:.4f} us/step  finite {a['outputs_finite']}")
PY
  grep "lstm alone layer 1\|placement\|timeline\|#   " $out/${tag}_prof.err
done
