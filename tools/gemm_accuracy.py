"""tools/gemm_accuracy.py -- how far is each GEMM flavour from the exact result?
fc1 -> bn1 -> tanh of one segment (hidden 1024, K = 2974) evaluated in float64 with numpy from the engine's own
input tap and the dequantised weights, against: the fp32-MFMA kernel, the bf16x3 kernel, the CPU oracle (fp32)."""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

pkg, po = ge.load_package(), ge.load_oracle()
H, N = 1024, 96 * 1024
d = tempfile.mkdtemp()
path = f"{d}/m.bin"
pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=7), H, compress=False)
hidden, targets = pkg.ggml.read_model(path)
wave = pkg.ggml.synth_audio(N, 321)
om = po.Model.load(path)
ref_out, taps = po.umx_inference(om, wave, n_buf=N, want_taps=True)
res = {}
for name, bx in (("f32 MFMA", False), ("bf16x3 MFMA", True)):
    eng = pkg.Engine.from_file(path, N, gemm="bf16x3" if bx else "f32")
    eng.infer_segment(wave, pkg.FLAG_DEBUG_TAPS)
    res[name] = ([eng.tap("fc1", t) for t in range(4)], eng.tap("x")[:, :2 * pkg.CROP].astype(np.float64),
                 [eng.tap("mask", t) for t in range(4)])
    eng.close()
x64 = res["f32 MFMA"][1]
print(f"{'target':6s} {'flavour':12s} {'max abs err':>12s} {'rel L2 err':>12s}   (fc1/bn1/tanh output vs float64)")
for t in range(4):
    tt = targets[t]
    g = lambda n: tt[n]["f32"].astype(np.float64)
    sc = np.tile(g("input_scale"), 2)
    mn = np.tile(g("input_mean"), 2)
    a = x64 * sc + mn
    y = a @ g("fc1.weight").reshape(H, -1).T
    y = (y - g("bn1.running_mean")) / np.sqrt(g("bn1.running_var") + 1e-5) * g("bn1.weight") + g("bn1.bias")
    y = np.tanh(y)
    for name in ("f32 MFMA", "bf16x3 MFMA"):
        e = res[name][0][t].astype(np.float64) - y
        print(f"{t:<6d} {name:12s} {np.abs(e).max():12.3e} {np.linalg.norm(e) / np.linalg.norm(y):12.3e}")
    e = taps["fc1_out"][t].astype(np.float64) - y
    print(f"{t:<6d} {'CPU oracle':12s} {np.abs(e).max():12.3e} {np.linalg.norm(e) / np.linalg.norm(y):12.3e}")
print("end-to-end mask, bf16x3 vs f32 MFMA: max abs",
      max(float(np.abs(res['bf16x3 MFMA'][2][t] - res['f32 MFMA'][2][t]).max()) for t in range(4)))
