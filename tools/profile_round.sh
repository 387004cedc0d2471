#!/bin/bash
# tools/profile_round.sh <tag>: the rocprofv3 evidence of one build, written under gpurun_out/prof_<tag>/
# (kernel trace of the pipelined bench; FETCH_SIZE and WRITE_SIZE in separate counter passes, serial segments)
tag=${1:-r02_v1}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-single-track --track-seconds 0 > $out/bench_under_rocprof.json 2> $out/trace.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-single-track --track-seconds 0 --serial > /dev/null 2> $out/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-single-track --track-seconds 0 --serial > /dev/null 2> $out/pmc_write.err
cd $R
python - <<PY
import csv, glob, collections, os
out="$out"
def agg(pattern, col):
    d=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob(out+"/"+pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name")==col:
                k=r["Kernel_Name"][:70]; d[k][0]+=1; d[k][1]+=float(r["Counter_Value"])
    return d
fe=agg("pmc_fetch/**/*counter_collection.csv","FETCH_SIZE"); wr=agg("pmc_write/**/*counter_collection.csv","WRITE_SIZE")
with open(out+"/pmc_fetch_write_per_kernel.csv","w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch(raw),WRITE_SIZE_KB_per_dispatch(raw)\n")
    for k in sorted(fe):
        n,v=fe[k]; w=wr.get(k,[1,0.0])
        f.write(f'"{k}",{n},{v/n:.1f},{w[1]/max(w[0],1):.1f}\n')
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    os.system(f"cp {f} {out}/kernel_stats.csv")
# keep the merge-back small
os.system(f"rm -rf {out}/trace {out}/pmc_fetch {out}/pmc_write")
PY
ls -la $out
