"""Stage-by-stage HIP-vs-oracle report (diagnostic; also imported by the gpu parity tests).
Usage on the GPU box: python tests/stagecheck.py [hidden] [frames] [flags]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def rel(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a.astype(np.complex128) - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def stage_report(hidden=128, n_buf=16 * 1024, n=None, flags=0, seed=5, segments=1, verbose=True):
    pkg, po = ge.load_package(), ge.load_oracle()
    n = n_buf if n is None else n
    with tempfile.TemporaryDirectory() as td:
        path = td + "/m.bin.gz"
        pkg.ggml.write_model(path, pkg.ggml.synth_weights(hidden, seed=seed), hidden)
        eng = pkg.Engine.from_file(path, segment_samples=n_buf)
        om = po.Model.load(path)
    state = po.stream_state(hidden)
    rep = {}
    for seg in range(segments):
        wave = pkg.ggml.synth_audio(n, seed=seed + 100 + seg)
        t0 = time.time()
        ref, taps = po.umx_inference(om, wave, n_buf=n_buf, state=state, flags=flags & 0xF01, want_taps=True)
        t_or = time.time() - t0
        t0 = time.time()
        got = eng.infer_segment(wave, flags | pkg.FLAG_DEBUG_TAPS)
        t_hip = time.time() - t0
        T = eng.T
        r = {}
        r["spec"] = rel(eng.tap("spec"), taps["spec"])
        r["mix_mag"] = rel(eng.tap("mix_mag"), taps["mix_mag"])
        r["x"] = rel(eng.tap("x")[:, :2 * pkg.CROP], taps["x"])
        for t in range(4):
            if flags & pkg.FLAG_SKIP_TARGET(t):
                continue
            r[f"fc1[{t}]"] = rel(eng.tap("fc1", t), taps["fc1_out"][t])
            r[f"lstm[{t}]"] = rel(eng.tap("lstm", t), taps["lstm_out"][t])
            r[f"fc2[{t}]"] = rel(eng.tap("fc2", t), taps["fc2_out"][t])
            r[f"mask[{t}]"] = rel(eng.tap("mask", t), taps["mask"][t])
            r[f"target_mag[{t}]"] = rel(eng.tap("target_mag", t), taps["target_mag"][t])
        for t in range(4):
            r[f"y[{t}]"] = rel(eng.tap("y", t), taps["y"][t])
            r[f"wave[{t}]"] = rel(got[t], ref[t])
            r[f"wave_maxabs[{t}]"] = float(np.abs(got[t] - ref[t]).max())
        r["state"] = rel(eng.stream_get(), state)
        r["persistent"] = eng.lstm_was_persistent()
        r["lstm_mode"] = eng.lstm_mode()
        r["t_oracle_s"], r["t_hip_s"] = t_or, t_hip
        rep[seg] = r
        if verbose:
            print(f"--- hidden={hidden} n_buf={n_buf} n={n} T={T} flags={flags:#x} segment {seg}")
            for k, v in r.items():
                print(f"   {k:18s} {v:.3e}" if isinstance(v, float) else f"   {k:18s} {v}")
            print("   stage ms:", {k: round(v, 3) for k, v in eng.stage_times().items()})
    eng.close()
    return rep


if __name__ == "__main__":
    hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
    segs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    stage_report(hidden, frames * 1024, flags=flags, segments=segs)
