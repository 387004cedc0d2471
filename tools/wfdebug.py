import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
torch.zeros(1).cuda()
import __graft_entry__ as ge
pkg = ge.load_package()
H, N = 1024, 24 * 1024
path = '/tmp/m_dbg.bin'
pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
waves = [pkg.ggml.synth_audio(N, 200 + i) for i in range(nseg)]
def taps(eng):
    d = {}
    for sl in range(min(nseg, 3)):
        for t in (0, 1, 2, 3):
            for nm in ('fc1','proj','lstm_l0','lstm_l1','lstm','target_mag'):
                d[f'{nm}[t{t}]@slot{sl}'] = eng.tap(f'{nm}@{sl}', t).copy()
    return d
e1 = pkg.Engine.from_file(path, N)
for i in range(nseg):
    out_s = e1.infer_segment(waves[i])
ts = taps(e1); st_s = e1.stream_get()
e2 = pkg.Engine.from_file(path, N)
ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
outs = [[torch.empty(2*N, dtype=torch.float32, device='cuda') for _ in range(4)] for _ in range(nseg)]
torch.cuda.synchronize()
for i in range(nseg):
    e2.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]])
e2.sync()
tp = taps(e2); st_p = e2.stream_get()
print(f"== nseg={nseg}: taps per slot (= segment), max abs diff pipelined vs serial; 'proj' = P after the LAST input projection (layer 2)")
for k in ts:
    dd = np.abs(ts[k] - tp[k])
    if dd.max() > 0:
        rows = np.nonzero(dd.reshape(dd.shape[0], -1).max(axis=1) > 0)[0]
        cols = np.nonzero(dd.reshape(dd.shape[0], -1).max(axis=0) > 0)[0]
        print(f"   {k:24s} max {dd.max():.3e} rows {rows[:6]}..{rows[-1]} ({len(rows)}) cols {cols[:6]}..{cols[-1]} ({len(cols)})")
print("   state diff per (target,layer):", np.abs(st_s-st_p).reshape(4,3,-1).max(axis=2).tolist())
for t in range(4):
    a = ts[f'lstm[t{t}]@slot1']; b = tp[f'lstm[t{t}]@slot1']
    d = np.abs(a-b)
    if d.max() == 0: continue
    print(f"target {t}: per-row max diff (fwd half):", ["%.1e" % v for v in d[:, :512].max(axis=1)[:8]], "... last", "%.1e" % d[-1, :512].max())
    print(f"           per-row max diff (bwd half):", ["%.1e" % v for v in d[:, 512:].max(axis=1)[:4]], "... last rows", ["%.1e" % v for v in d[-4:, 512:].max(axis=1)])
    r0 = np.nonzero(d[0, :512] > 0)[0]
    print(f"           row 0 fwd units differing: {len(r0)} first {r0[:12]}")
