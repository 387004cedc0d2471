"""tools/corun_probe.py -- how much does a co-running kernel of a given kind slow the persistent LSTM kernel?
Queues a long run of torch work on a side stream (MFMA-bound fp32 GEMMs, an HBM-bound copy, or nothing) and
times one serial engine segment meanwhile; prints the LSTM stage times."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
H, N = 1024, pkg.SEGMENT_SAMPLES
torch.zeros(1).cuda()
d = tempfile.mkdtemp()
wpath = f"{d}/w.bin"
pkg.ggml.write_model(wpath, pkg.ggml.synth_weights(H, seed=1), H, compress=False)
eng = pkg.Engine.from_file(wpath, N)
wave = pkg.ggml.synth_audio(N, 1)
a_dev = torch.from_numpy(np.ascontiguousarray(wave.T).ravel()).cuda()
outs = [torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)]


def segment():
    eng.infer_segment_device(a_dev.data_ptr(), N, [o.data_ptr() for o in outs])
    eng.sync()
    return eng.stage_times()


segment()
side = torch.cuda.Stream()
A = torch.randn(8192, 8192, device="cuda")
B = torch.randn(8192, 8192, device="cuda")
C = torch.empty(8192, 8192, device="cuda")
big1 = torch.empty(1 << 28, device="cuda")  # 1 GiB
big2 = torch.empty(1 << 28, device="cuda")
small = torch.randn(1 << 20, device="cuda")
torch.backends.cuda.matmul.allow_tf32 = False


def run(kind, reps):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(reps):
            if kind == "gemm":
                torch.mm(A, B, out=C)
            elif kind == "copy":
                big2.copy_(big1)
            elif kind == "alu":
                for _ in range(20):
                    small.sin_()
    t0 = time.perf_counter()
    st = segment()
    dt = (time.perf_counter() - t0) * 1e3
    busy = not side.query()
    torch.cuda.synchronize()
    print(f"{kind:5s}: segment {dt:6.2f} ms  rec0 {st['lstm_rec0']:.2f} rec1 {st['lstm_rec1']:.2f} rec2 {st['lstm_rec2']:.2f}  "
          f"ih1 {st['lstm_ih1']:.2f} fc1 {st['fc1']:.2f} wiener {st['wiener']:.2f}  (side stream still busy at the end: {busy})")


for kind, reps in (("none", 0), ("gemm", 12), ("copy", 120), ("alu", 400), ("none", 0)):
    run(kind, reps)
eng.close()
