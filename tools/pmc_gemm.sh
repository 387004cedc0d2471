#!/bin/bash
# tools/pmc_gemm.sh [bench args]: a few SQ counters for the GEMM / LSTM kernels (separate passes), serial steps
out=$PWD/gpurun_out/pmc_gemm; mkdir -p $out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $out/$tag -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie --no-single-track --track-seconds 0 --serial "$@" > /dev/null 2> $out/$tag.err
done
python - <<PY
import csv, glob, collections
d=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemm" not in k and "lstm_batch" not in k and "split" not in k: continue
        k=k[:52]
        e=d[k][r["Counter_Name"]]; e[0]+=1; e[1]+=float(r["Counter_Value"])
for k in sorted(d):
    print(k, {c: round(v[1]/v[0]) for c,v in d[k].items()})
PY
rm -rf $out/*/
