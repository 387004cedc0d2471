import sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
pkg = ge.load_package()
H, N = 1024, 40 * 1024
d = tempfile.mkdtemp()
p = f"{d}/m.bin"
pkg.ggml.write_model(p, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
eng = pkg.Engine.from_file(p, N, gemm="bf16x3")
w = pkg.ggml.synth_audio(N, 200)
res = []
for i in range(4):
    eng.stream_reset()
    res.append(eng.infer_segment(w, pkg.FLAG_DEBUG_TAPS))
    taps = {k: eng.tap(k, 0) for k in ("fc1", "lstm_l0", "proj", "mask")}
    if i == 0:
        t0 = taps
    else:
        print(i, {k: float(np.abs(taps[k] - t0[k]).max()) for k in taps}, max(float(np.abs(res[i][t] - res[0][t]).max()) for t in range(4)))
