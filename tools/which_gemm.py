#!/usr/bin/env python
"""tools/which_gemm.py [tracks]: which GEMM kernel the engine launches per stage at full size, and the stand-alone stage times."""
import sys, os, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = pkg.SEGMENT_SAMPLES
d = tempfile.mkdtemp()
path = os.path.join(d, "m.bin")
pkg.ggml.write_model(path, pkg.ggml.synth_weights(1024, seed=0), 1024, compress=False)
eng = pkg.Engine.from_file(path, N, tracks=B)
w = pkg.ggml.synth_audio(N, 1)
for i in range(3):
    eng.infer_batch([w] * B)
print("gemm kernels:", [eng.gemm_kernel_name(m) for m in range(4)], "lstm:", eng.lstm_kernel_name())
print({k: round(v, 3) for k, v in eng.stage_times().items()})
print({k: round(v, 3) for k, v in eng.stage_kernel_times().items()})
eng.close()
