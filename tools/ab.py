#!/usr/bin/env python
"""A/B runner for kernel variants on the GPU box: tools/ab.py <outdir> <tracks,tracks,..> <variant .so | 'default' | env:NAME=VALUE[,NAME=VALUE]> ...
Prints one compact line per (variant, tracks): ms/step, x realtime, stand-alone stage times, LSTM phase cycles."""
import json
import os
import subprocess
import sys

out, tracks, variants = sys.argv[1], [int(x) for x in sys.argv[2].split(",")], sys.argv[3:]
extra = os.environ.get("AB_BENCH_ARGS", "").split()
os.makedirs(out, exist_ok=True)
for v in variants:
    for b in tracks:
        env = dict(os.environ)
        if v.startswith("env:"):
            env.update(kv.split("=", 1) for kv in v[4:].split(","))
        elif v != "default":
            env["UMX_HIP_LIB"] = os.path.abspath(v)
        tag = os.path.basename(v).replace("libumx_hip_", "").replace(".so", "").replace("env:", "").replace("=", "")
        cmd = [sys.executable, "bench.py", "--tracks", str(b), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-pcie", "--no-single-track", "--track-seconds", "0"] + extra
        if os.environ.get("AB_PROFILE"):  # the in-kernel profiler slows the profiled workgroup, and with it its whole chain
            cmd.append("--lstm-profile")
        if b == 1:
            cmd.append("--batched-lstm")
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        open(f"{out}/{tag}_B{b}.json", "w").write(p.stdout)
        open(f"{out}/{tag}_B{b}.err", "w").write(p.stderr)
        try:
            j = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print(tag, b, "FAILED", p.stderr[-400:])
            continue
        al = j["stages_ms_unpipelined"]
        lstm = sum(al[f"lstm_rec{l}"] for l in range(3)) / 3
        gemm = al["fc1"] + al["fc2"] + al["fc3_mask"] + sum(al[f"lstm_ih{l}"] for l in range(3))
        prof = [ln for ln in p.stderr.splitlines() if ln.startswith("# lstm alone layer 1 wave")]
        print(f"{tag:12s} B={b:2d} {j['ms_per_step']:8.3f} ms/step {j['value']:9.1f}x  alone: lstm/launch {lstm:7.3f} ms "
              f"({lstm * 1e3 / j['config']['frames']:.3f} us/step) gemm {gemm:7.3f} serial {j['ms_per_step_unpipelined']:8.3f}  "
              f"fc1 {al['fc1']:.2f} ih {al['lstm_ih0']:.2f}/{al['lstm_ih1']:.2f}/{al['lstm_ih2']:.2f} fc2 {al['fc2']:.2f} fc3 {al['fc3_mask']:.2f} "
              f"chk {j.get('checked_max_abs')}", flush=True)
        for ln in prof:
            print("      ", ln[2:], flush=True)
