"""tests/sdr_check.py -- real-weight validation helper; test infrastructure; exercised by tests/test_gpu_sdr.py on synthetic
UMX-L-shaped weights.

    python tests/sdr_check.py <ggml model (.bin or .bin.gz)> <mix.wav> [<dir with target_{0..3}.wav>] [--seconds S]

Separates the mix with the MI355X engine (umx_hip_shift_inference, offset 4033 = the reference's unseeded
rand() % 22050) and with the CPU oracle (same offset), and prints a museval-free SDR per stem,
10 log10(|ref|^2 / |ref - est|^2): engine vs oracle, and -- if a directory of reference stems is given, e.g. the
output of the real umx.cpp binary or of the PyTorch model -- engine vs those.  With the real
ggml-model-umxl-u8.bin.gz (sha256 6a013ecf...) this makes the README's SDR table checkable; without it, synthetic
weights exercise the same path.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

NAMES = ["bass", "drums", "other", "vocals"]  # target_0..3 (umx.cpp:75-96)


def sdr(ref, est):
    ref, est = np.asarray(ref, np.float64), np.asarray(est, np.float64)
    return 10 * np.log10(max(np.sum(ref ** 2), 1e-30) / max(np.sum((ref - est) ** 2), 1e-30))


def run(model, wav, stems_dir=None, seconds=0.0, write_oracle_stems=None, out=print):
    """-> {"engine_vs_oracle": [4 SDRs], "engine_vs_dir": [4 SDRs] or None}.  write_oracle_stems: a directory that receives
    the oracle's stems as target_{0..3}.wav (the files the real umx.cpp binary would have written, umx.cpp:75-96)."""
    pkg, po = ge.load_package(), ge.load_oracle()
    wave, _channels = pkg.wav_load(wav)
    if seconds > 0:
        wave = wave[:, :int(seconds * 44100)]
    N = pkg.SEGMENT_SAMPLES
    eng = pkg.Engine.from_file(model, N)
    got = eng.separate(wave, shift_offset=4033)
    om = po.Model.load(model)
    ref = po.shift_inference(om, wave, N, offset=4033)
    if write_oracle_stems:
        Path(write_oracle_stems).mkdir(parents=True, exist_ok=True)
        for t in range(4):
            pkg.wav_write(Path(write_oracle_stems) / f"target_{t}.wav", ref[t])
    res = {"engine_vs_oracle": [], "engine_vs_dir": [] if stems_dir else None}
    out(f"{wave.shape[1] / 44100:.1f} s, hidden {eng.hidden}; SDR in dB")
    for t in range(4):
        res["engine_vs_oracle"].append(sdr(ref[t], got[t]))
        line = f"  target_{t} ({NAMES[t]:6s}) engine vs oracle {res['engine_vs_oracle'][-1]:7.2f}"
        if stems_dir:
            st, _ = pkg.wav_load(str(Path(stems_dir) / f"target_{t}.wav"))
            n = min(st.shape[1], got[t].shape[1])
            res["engine_vs_dir"].append(sdr(st[:, :n], got[t][:, :n]))
            line += f"   engine vs {stems_dir}/target_{t}.wav {res['engine_vs_dir'][-1]:7.2f}"
        out(line)
    eng.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("wav")
    ap.add_argument("stems_dir", nargs="?")
    ap.add_argument("--seconds", type=float, default=0.0, help="only the first S seconds (the oracle is slow)")
    ap.add_argument("--write-oracle-stems", default=None, help="directory that receives the oracle's stems as wav files")
    a = ap.parse_args()
    run(a.model, a.wav, a.stems_dir, a.seconds, a.write_oracle_stems)


if __name__ == "__main__":
    main()
