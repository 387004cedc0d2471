"""tools/wiener_ab.py -- fused (wiener_istft.h) vs unfused Wiener / inverse-STFT path against the oracle on one segment:
relative L2 error of the y tap and of the stems, and which bins / frames carry the difference."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

pkg, po = ge.load_package(), ge.load_oracle()
H, N = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 1024
d = tempfile.mkdtemp()
path = f"{d}/m.bin"
pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=17), H, compress=False)
wave = pkg.ggml.synth_audio(N, 5)
om = po.Model.load(path)
ref, taps = po.umx_inference(om, wave, n_buf=N, want_taps=True)


def run(mode):
    os.environ["UMX_WIENER"] = mode
    eng = pkg.Engine.from_file(path, N)
    out = eng.infer_segment(wave, pkg.FLAG_DEBUG_TAPS)
    y = [eng.tap("y", t) for t in range(4)]
    eng.close()
    return out, y


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


res = {m: run(m) for m in ("unfused", "fused")}
for m, (out, y) in res.items():
    print(m, "y vs oracle", [f"{rel(y[t], taps['y'][t]):.2e}" for t in range(4)], "wave vs oracle", [f"{rel(out[t], ref[t]):.2e}" for t in range(4)])
print("fused vs unfused y", [f"{rel(res['fused'][1][t], res['unfused'][1][t]):.2e}" for t in range(4)],
      "wave", [f"{rel(res['fused'][0][t], res['unfused'][0][t]):.2e}" for t in range(4)])
yf, yu = np.asarray(res["fused"][1][0], np.float64), np.asarray(res["unfused"][1][0], np.float64)
dd = np.abs(yf - yu)
print("y shape", yf.shape, "worst element", np.unravel_index(np.argmax(dd), dd.shape), dd.max(), "per-bin max (top 5 bins):",
      np.argsort(dd.reshape(-1, dd.shape[-1]).max(0) if dd.ndim >= 2 else dd)[-5:])
