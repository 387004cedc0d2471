#!/bin/bash
# tools/trace_overlap.sh <tag> [bench args]: kernel-trace timeline of a short run -> which kernel families overlap
tag=$1; shift
out=$PWD/gpurun_out/trace_$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-single-track "$@" > $out/bench.json 2> $out/err.log
cd $R
python - <<PY
import csv, glob, collections
f=glob.glob("$out/t/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
def fam(n):
    for k in ("gemm_planes","lstm_batch","lstm_persistent","split_planes","wiener_stats","wiener_apply","wiener_finish","istft_frames","istft_ola","stft_kernel"):
        if k in n: return k
    return None
ev=[]
for r in rows:
    k=fam(r["Kernel_Name"])
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
# take the steady-state window: the last 60 % of the span
t0=ev[0][0]; t1=max(e[1] for e in ev); lo=t0+(t1-t0)*0.4
ev=[e for e in ev if e[0]>=lo]
span=max(e[1] for e in ev)-min(e[0] for e in ev)
busy=collections.Counter(); tot=0
pts=sorted(set([e[0] for e in ev]+[e[1] for e in ev]))
import bisect
# sweep
active=collections.Counter(); idx=0
evs=sorted([(e[0],1,e[2]) for e in ev]+[(e[1],-1,e[2]) for e in ev])
last=evs[0][0]; union=0; multi=0; pair=collections.Counter()
for t,d,k in evs:
    dt=t-last
    if dt>0:
        act=[a for a,c in active.items() if c>0]
        if act: union+=dt
        if len(act)>1:
            multi+=dt
            pair[tuple(sorted(act))]+=dt
        for a in act: busy[a]+=dt
    active[k]+=d; last=t
print(f"window {span/1e6:.2f} ms, some kernel active {union/1e6:.2f} ms ({100*union/span:.1f} %), >=2 families active {multi/1e6:.2f} ms ({100*multi/span:.1f} %)")
for k,v in busy.most_common(): print(f"  {k:16s} active {v/1e6:8.2f} ms")
for k,v in pair.most_common(8): print(f"  overlap {k}: {v/1e6:.2f} ms")
PY
rm -rf $out/t
