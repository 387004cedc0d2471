#!/usr/bin/env python
"""tools/stage_ab.py <outdir> <tracks> <variant .so | default> ...: the stand-alone time of every stage of a step (one call at a
time) per variant library, one line each -- for kernels tools/ab.py does not print (Wiener statistics, fused Wiener / inverse STFT)."""
import json, os, subprocess, sys
out, tracks, variants = sys.argv[1], sys.argv[2], sys.argv[3:]
os.makedirs(out, exist_ok=True)
for v in variants:
    env = dict(os.environ)
    if v != "default":
        env["UMX_HIP_LIB"] = os.path.abspath(v)
    tag = os.path.basename(v).replace("libumx_hip_", "").replace(".so", "")
    p = subprocess.run([sys.executable, "bench.py", "--tracks", tracks, "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pcie", "--no-single-track", "--track-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    open(f"{out}/{tag}.json", "w").write(p.stdout)
    open(f"{out}/{tag}.err", "w").write(p.stderr)
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        print(tag, "FAILED", p.stderr[-400:]); continue
    al = j["stages_ms_unpipelined"]
    print(f"{tag:14s} step {j['ms_per_step']:7.3f} ms (min {j.get('ms_per_step_min')})  " + " ".join(f"{k} {v:.3f}" for k, v in al.items()), flush=True)
