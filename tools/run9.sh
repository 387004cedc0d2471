cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "full_size_segment or stage_parity_small_persistent or flags_no_wiener" 2>&1 | tail -4
python tools/ab.py gpurun_out/r2k 16 default
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r2k/default_B16.json").read().strip().splitlines()[-1])
B=16
print({k: round(v/B,3) for k,v in j["stages_ms_unpipelined"].items()})
PY
