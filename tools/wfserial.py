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
ref = None
for rep in range(4):
    e1 = pkg.Engine.from_file(path, N)
    serial = [e1.infer_segment(w, 0) for w in waves]
    st = e1.stream_get(); e1.close()
    if ref is None:
        ref = (serial, st); continue
    diffs = [max(float(np.abs(serial[i][t] - ref[0][i][t]).max()) for t in range(4)) for i in range(nseg)]
    print("serial rep", rep, "vs rep 0:", ['%.1e' % d for d in diffs], "state", np.abs(st-ref[1]).max())
# pipelined where every segment is followed by a sync through the DEVICE api (masks 1,2,4 only)
e2 = pkg.Engine.from_file(path, N)
ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
outs = [[torch.empty(2*N, dtype=torch.float32, device='cuda') for _ in range(4)] for _ in range(nseg)]
torch.cuda.synchronize()
for i in range(nseg):
    e2.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], 0)
    e2.sync()
diffs = [max(float(np.abs(outs[i][t].cpu().numpy().reshape(N,2).T - ref[0][i][t]).max()) for t in range(4)) for i in range(nseg)]
print("device api + sync each:", ['%.1e' % d for d in diffs])
# two segments back to back then sync (masks 1,3,6,4), repeated
for rep in range(3):
    e2 = pkg.Engine.from_file(path, N)
    for i in range(0, nseg, 2):
        e2.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], 0)
        e2.infer_segment_device(ins[i+1].data_ptr(), N, [o.data_ptr() for o in outs[i+1]], 0)
        e2.sync()
    diffs = [max(float(np.abs(outs[i][t].cpu().numpy().reshape(N,2).T - ref[0][i][t]).max()) for t in range(4)) for i in range(nseg)]
    print("pairs back-to-back:", ['%.1e' % d for d in diffs])
    e2.close()
