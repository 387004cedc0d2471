#!/usr/bin/env python
"""tools/one_track_stages.py: the one-track (latency) engine -- stand-alone stage times of a lone 60 s segment and its wall time."""
import sys, os, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
N = pkg.SEGMENT_SAMPLES
d = tempfile.mkdtemp()
path = os.path.join(d, "m.bin")
pkg.ggml.write_model(path, pkg.ggml.synth_weights(1024, seed=0), 1024, compress=False)
for gemm in (None, "planes"):
    eng = pkg.Engine.from_file(path, N, gemm=gemm)
    w = np.ascontiguousarray(pkg.ggml.synth_audio(N, 1).T).ravel()
    a = torch.from_numpy(w).cuda()
    outs = [torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)]
    ptrs = [o.data_ptr() for o in outs]
    lone = []
    for i in range(8):
        t0 = time.perf_counter()
        eng.infer_segment_device(a.data_ptr(), N, ptrs, 0)
        eng.sync()
        lone.append((time.perf_counter() - t0) * 1e3)
    st = eng.stage_times()
    print("gemm", gemm or "bf16x3 (default)", "lone segment ms (median of 6):", round(float(np.median(lone[2:])), 3), "kernels:", [eng.gemm_kernel_name(m) for m in range(4)], eng.lstm_kernel_name())
    print("  ", {k: round(v, 3) for k, v in st.items()}, "sum", round(sum(st.values()), 3))
    eng.close()
