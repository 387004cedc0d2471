import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
torch.zeros(1).cuda()
import __graft_entry__ as ge
pkg = ge.load_package()
H, N = 1024, 24 * 1024
path = '/tmp/m_dbg.bin'
pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
nseg = 6
waves = [pkg.ggml.synth_audio(N, 200 + i) for i in range(nseg)]
for mode in ("wavefront",):
    os.environ["UMX_PIPELINE"] = mode
    for flags, name in ((0, "fast"),):
        e1 = pkg.Engine.from_file(path, N)
        serial = [e1.infer_segment(w, flags) for w in waves]
        st_s = e1.stream_get(); e1.close()
        bad = 0
        for rep in range(3):
            e2 = pkg.Engine.from_file(path, N)
            ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
            outs = [[torch.empty(2*N, dtype=torch.float32, device='cuda') for _ in range(4)] for _ in range(nseg)]
            torch.cuda.synchronize()
            for i in range(nseg):
                e2.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], flags)
            e2.sync()
            mode_used = e2.lstm_mode()
            diffs = [max(float(np.abs(outs[i][t].cpu().numpy().reshape(N,2).T - serial[i][t]).max()) for t in range(4)) for i in range(nseg)]
            st_p = e2.stream_get(); e2.close()
            bad += any(d > 0 for d in diffs)
            print(f"{mode:9s} {name}: lstm_mode {mode_used} per-segment max diff {['%.1e' % d for d in diffs]} state diff {np.abs(st_s-st_p).max():.1e}")
