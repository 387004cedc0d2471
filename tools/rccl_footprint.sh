#!/bin/bash
# tools/rccl_footprint.sh: what RCCL's send / recv kernels occupy beside the engine's grids -- one 600 s track through the multi-GPU driver
# on ONE GPU in loopback (every hop a grouped self send + receive), --mode track and --mode targets, under rocprofv3 --kernel-trace:
# grid and workgroup size, count and duration of every RCCL kernel -> gpurun_out/rccl_footprint.txt
out=$PWD/gpurun_out/rccl_fp; mkdir -p $out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for mode in track targets; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/$mode -- python $R/bench.py --mode $mode --loopback --steps 2 --warmup 1 > $out/$mode.json 2> $out/$mode.err
done
cd $R
python - <<PY > gpurun_out/rccl_footprint.txt
import csv, glob, collections, json
for mode in ("track", "targets"):
    rows = []
    for f in glob.glob("$out/%s/**/*kernel_trace.csv" % mode, recursive=True):
        rows += list(csv.DictReader(open(f)))
    try:
        line = json.loads([l for l in open("$out/%s.json" % mode).read().splitlines() if l.startswith("{")][-1])
        print("# --mode %s --loopback: %.1f ms per 600 s track; rccl stats %s" % (mode, line["ms_per_step"], line["config"].get("rccl")))
    except Exception as e:
        print("# --mode %s: no bench line (%r)" % (mode, e))
    if rows:
        print("# columns of the trace:", list(rows[0].keys()))
    agg = collections.defaultdict(lambda: [0, 0.0, set()])
    for r in rows:
        name = r["Kernel_Name"]
        key = name[:60]
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[2].add((r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
    for k, (n, us, shapes) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-62s calls %5d  total %10.1f us  mean %8.1f us  (grid, workgroup) %s" % (k, n, us, us / n, sorted(shapes)[:4]))
PY
rm -rf $out/track $out/targets
cat gpurun_out/rccl_footprint.txt
