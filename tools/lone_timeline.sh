#!/bin/bash
# tools/lone_timeline.sh: kernel timeline (rocprofv3 --kernel-trace) of the LAST lone segment of tools/one_track_stages.py's default context:
# start offset, duration, queue and grid of every kernel -> which kernels really overlap in the target-stream mode (engine_stages.h)
out=$PWD/gpurun_out/lone_tl; mkdir -p $out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python $R/tools/one_track_stages.py > $out/run.log 2>&1
cd $R
python - <<PY
import csv, glob
rows=[]
for f in glob.glob("$out/t/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows=[r for r in rows if "umx::" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# segments of the first engine: 8 lone segments; a segment starts with stft_kernel.  take the 8th
starts=[i for i,r in enumerate(rows) if "stft_kernel" in r["Kernel_Name"]]
i0=starts[7]; i1=starts[8] if len(starts)>8 else len(rows)
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    nm=r["Kernel_Name"].replace("void ","").split("(")[0][:44]
    print("%-46s q %-3s start %8.1f us  dur %8.1f us  grid %s" % (nm, r["Queue_Id"], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"]))
PY
rm -rf $out/t
