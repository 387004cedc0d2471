#!/bin/bash
# tools/prof_kernels.sh <tag> [bench args]: rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/prof_<tag>/kernel_stats.csv
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pcie --no-single-track "$@" > $out/bench_under_rocprof.json 2> $out/trace.err
cd $R
for f in $(find $out/trace -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
rm -rf $out/trace
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:24]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} {r["Percentage"]}%')
PY
