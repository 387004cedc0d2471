import sys, tempfile, os, ctypes as C
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
pkg = ge.load_package()
torch.zeros(1).cuda()
H, N, NSEG = 1024, 40 * 1024, 8
d = tempfile.mkdtemp()
p = f"{d}/m.bin"
pkg.ggml.write_model(p, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
eng = pkg.Engine.from_file(p, N, gemm="bf16x3" if os.environ.get("BX", "1") == "1" else "planes")
lib = eng.lib
lib.umx_hip_debug_lds_guard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint)]
waves = [pkg.ggml.synth_audio(N, 200 + i) for i in range(NSEG)]
ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
outs = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in range(NSEG)]
torch.cuda.synchronize()
for rep in range(3):
    lib.umx_hip_debug_lds_guard(eng.h, 40, 200, None)
    for i in range(NSEG):
        eng.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], int(os.environ.get('FL','0'),0))
    eng.sync()
    out = (C.c_uint * 2)()
    lib.umx_hip_debug_lds_guard(eng.h, 0, 0, out)
    print("rep", rep, "corrupted LDS words", out[0], "events", out[1])
