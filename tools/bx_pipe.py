import sys, tempfile, os
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
pkg = ge.load_package()
torch.zeros(1).cuda()
H, N, NSEG = 1024, int(os.environ.get("NS", 40 * 1024)), int(os.environ.get("NSEG", 4))
d = tempfile.mkdtemp()
p = f"{d}/m.bin"
pkg.ggml.write_model(p, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
import os
eng = pkg.Engine.from_file(p, N, gemm="bf16x3" if os.environ.get("BX", "1") == "1" else "planes")
waves = [pkg.ggml.synth_audio(N, 200 + i) for i in range(NSEG)]
eng.stream_reset()
serial = [eng.infer_segment(w, int(os.environ.get('FL', '0'), 0)) for w in waves]
ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
def pipe():
    eng.stream_reset()
    outs = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in range(NSEG)]
    torch.cuda.synchronize()
    for i in range(NSEG):
        eng.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], int(os.environ.get('FL', '0'), 0))
    eng.sync()
    return [[o.cpu().numpy().reshape(N, 2).T for o in oo] for oo in outs]
a, b = pipe(), pipe()
for i in range(NSEG):
    print(i, "pipe-vs-pipe", max(float(np.abs(a[i][t] - b[i][t]).max()) for t in range(4)),
          "pipe-vs-serial", max(float(np.abs(a[i][t] - serial[i][t]).max()) for t in range(4)),
          "scale", float(np.abs(serial[i][0]).max()))

def pipe_taps():
    eng.stream_reset()
    outs = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in range(NSEG)]
    torch.cuda.synchronize()
    for i in range(NSEG):
        eng.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], pkg.FLAG_DEBUG_TAPS | int(os.environ.get('FL', '0'), 0))
    eng.sync()
    return {k: [np.abs(eng.tap(k, t)) if k == "spec" else eng.tap(k, t) for t in range(4)] for k in ("spec", "mix_mag", "x", "fc1", "lstm_l0", "lstm_l1", "lstm", "proj", "mask", "target_mag")}
ta, tb = pipe_taps(), pipe_taps()
for k in ta:
    print(k, [float(np.abs(ta[k][t] - tb[k][t]).max()) for t in range(4)])
xa, xb = ta["x"][0], tb["x"][0]
print("x shape", xa.shape)
dif = np.argwhere(np.abs(xa - xb) > 0)
print("n diff", len(dif), "rows", np.unique(dif[:, 0])[:20] if dif.ndim == 2 else dif[:20], "cols", np.unique(dif[:, 1])[:20] if dif.ndim == 2 else None)
ref_ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()) for w in waves]
for i in range(NSEG):
    print("input", i, "changed:", bool((ins[i].cpu() != ref_ins[i]).any().item()))
