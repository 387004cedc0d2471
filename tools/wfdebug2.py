import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
torch.zeros(1).cuda()
import __graft_entry__ as ge
pkg = ge.load_package()
H, N = 1024, 24 * 1024
path = '/tmp/m_dbg.bin'
pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
nseg = 3
waves = [pkg.ggml.synth_audio(N, 200 + i) for i in range(nseg)]
def taps(eng, sl):
    d = {}
    d['spec'] = eng.tap(f'spec@{sl}').copy(); d['x'] = eng.tap(f'x@{sl}').copy()
    for t in range(4):
        for nm in ('fc1','proj','lstm_l0','lstm_l1','lstm','target_mag'):
            d[f'{nm}[t{t}]'] = eng.tap(f'{nm}@{sl}', t).copy()
    return d
e1 = pkg.Engine.from_file(path, N)
T1 = []
for i in range(nseg):
    e1.infer_segment(waves[i]); T1.append(taps(e1, i % 3))
e2 = pkg.Engine.from_file(path, N)
ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
outs = [[torch.empty(2*N, dtype=torch.float32, device='cuda') for _ in range(4)] for _ in range(nseg)]
torch.cuda.synchronize()
T2 = []
for i in range(nseg):
    e2.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], 0)
    e2.sync(); T2.append(taps(e2, i % 3))
for i in range(nseg):
    for k in T1[i]:
        dd = np.abs(T1[i][k] - T2[i][k]).max()
        if dd > 0: print(f"seg {i} {k:16s} {dd:.3e}")
print("done")
