cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
( time timeout 2000 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r2h/gputests.log 2>&1
cat gpurun_out/r2h/gputests.log
( time python bench.py ) > gpurun_out/r2h/bench_default.json 2> gpurun_out/r2h/bench_default.err
tail -3 gpurun_out/r2h/bench_default.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r2h/bench_default.json").read().strip().splitlines()[0])
for k in ("value","ms_per_step","value_pcie","ms_per_step_pcie","single_track","roofline","gemm_view"):
    print(k, j.get(k))
print("cpu", j.get("cpu_baseline"))
for k in j["kernels"]:
    print({a:k[a] for a in ("kernel","launches_per_step","launch_ms","launch_ms_alone","frac","frac_alone")})
PY
