"""tests/sdr_check.py -- real-weight validation helper; test infrastructure, not collected by pytest.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("wav")
    ap.add_argument("stems_dir", nargs="?")
    ap.add_argument("--seconds", type=float, default=0.0, help="only the first S seconds (the oracle is slow)")
    a = ap.parse_args()
    pkg, po = ge.load_package(), ge.load_oracle()
    wave, _channels = pkg.wav_load(a.wav)
    if a.seconds > 0:
        wave = wave[:, :int(a.seconds * 44100)]
    N = pkg.SEGMENT_SAMPLES
    eng = pkg.Engine.from_file(a.model, N)
    got = eng.separate(wave, shift_offset=4033)
    om = po.Model.load(a.model)
    ref = po.shift_inference(om, wave, N, offset=4033)
    print(f"{wave.shape[1] / 44100:.1f} s, hidden {eng.hidden}; SDR in dB")
    for t in range(4):
        line = f"  target_{t} ({NAMES[t]:6s}) engine vs oracle {sdr(ref[t], got[t]):7.2f}"
        if a.stems_dir:
            st, _ = pkg.wav_load(str(Path(a.stems_dir) / f"target_{t}.wav"))
            n = min(st.shape[1], got[t].shape[1])
            line += f"   engine vs {a.stems_dir}/target_{t}.wav {sdr(st[:, :n], got[t][:, :n]):7.2f}"
        print(line)
    eng.close()


if __name__ == "__main__":
    main()
