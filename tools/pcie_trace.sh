#!/bin/bash
# tools/pcie_trace.sh <tag> [bench args]: kernel + memory-copy timeline of the PCIe-inclusive leg of bench.py (value_pcie), reduced
# on the box to (a) per-copy rate and (b) a merged timeline of the last steps, under gpurun_out/pcie_<tag>/
tag=${1:-r03}; shift
out=$PWD/gpurun_out/pcie_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-single-track "$@" > $out/bench.json 2> $out/trace.err
cd $R
python - <<PY
import csv, glob
out="$out"
cp=[]; ks=[]
for f in glob.glob(out+"/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        b = r.get("Bytes") or r.get("Size") or r.get("Bytes_Copied") or "0"
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        cp.append((s_, e_, r.get("Direction",""), int(float(b)) if float(b) > 0 else (21168000 if e_ - s_ > 100000 else 0)))
    hdr = open(f).readline().strip()
for f in glob.glob(out+"/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id","")))
cp.sort(); ks.sort()
big=[c for c in cp if c[3] > 1<<20]
with open(out+"/summary.txt","w") as w:
    w.write("memory_copy_trace columns: " + (hdr if cp else "-") + "\n")
    w.write(f"copies: {len(cp)} total, {len(big)} above 1 MB; kernels: {len(ks)}\n")
    for d in sorted({c[2] for c in big}):
        sel=[c for c in big if c[2]==d]
        tot=sum(c[3] for c in sel); dur=sum(c[1]-c[0] for c in sel)
        w.write(f"{d}: {len(sel)} copies, {tot/1e9:.2f} GB, sum of durations {dur/1e6:.1f} ms -> {tot/max(dur,1):.2f} GB/s while a copy runs; "
                f"median copy {sorted(c[1]-c[0] for c in sel)[len(sel)//2]/1e3:.0f} us\n")
    if big:
        t_end=big[-1][1]; t0=t_end-400_000_000
        ev=[(s,e,"COPY "+d+f" {b/1e6:.0f}MB","") for s,e,d,b in cp if e>t0 and b>1<<20]+[(s,e,n,q) for s,e,n,q in ks if e>t0 and s<t_end]
        ev.sort()
        # merge runs of the same label
        merged=[]
        for s,e,n,q in ev:
            if merged and merged[-1][2]==n and s-merged[-1][1] < 2_000_000:
                merged[-1]=(merged[-1][0], max(e,merged[-1][1]), n, merged[-1][3]+1)
            else:
                merged.append((s,e,n,1))
        w.write("\nlast 400 ms, merged runs: start_ms end_ms count label\n")
        for s,e,n,c in merged:
            w.write(f"{(s-t0)/1e6:9.3f} {(e-t0)/1e6:9.3f} {c:4d} {n}\n")
import os
os.system(f"rm -rf {out}/trace")
PY
ls -la $out
