"""tools/gemm_accuracy.py -- how far is each arithmetic flavour from the exact result?  (The bench line says dtype "f32":
this is the evidence that the bf16-split matrix-core paths are no less accurate than fp32 arithmetic.)

Evaluated in float64 from the engine's OWN taps (so that only the stage under test contributes) and the weights the
reference would use (fl(q*scale+offset), model.cpp:610-616):
  fc1 -> bn1 -> tanh       K = 2974, u8 weights      [planes: exact one-plane weights, 2 plane products + the row-sum term, affine map on the sum]
  fc2 -> bn2 -> relu       K = 2048, u16 weights     [planes: two weight planes whose sum is the file's integer, 3 of the 4 plane products + the row-sum term]
  3-layer BiLSTM           2584-step-class recurrence, u8 W_hh, from the engine's fc1 output (torch float64 LSTM)
for the GEMM flavours planes / bf16x3 (staged split), the single-track (VALU) and the batched (matrix-core) LSTM kernels, and the
CPU oracle (fp32).  "64 lanes (shipped)" is the configuration the bench times: the launches are large enough for the persistent 256 x 256
plane GEMM (csrc/gemm_planes_ps.h) and run the recurrence in workgroups of 8 lanes x 64 units, two octets in turn (lstm_batch8_kernel, csrc/lstm_batch8.h);
the one- and two-lane rows run the 128 x 128 lock-step tiles and the same recurrence kernel (same arithmetic per element, tested bitwise).  (The fp32-MFMA flavour of rounds 1-2 is gone; its figures are in profiles/r02_accuracy_vs_float64.txt.)"""
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

pkg, po = ge.load_package(), ge.load_oracle()
H, N = 1024, 128 * 1024  # 64 lanes x 129 frames = 8256 rows: more 256 x 256 tiles than CUs in every GEMM
d = tempfile.mkdtemp()
path = f"{d}/m.bin"
pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=7), H, compress=False)
hidden, targets = pkg.ggml.read_model(path)
wave = pkg.ggml.synth_audio(N, 321)
om = po.Model.load(path)
ref_out, taps = po.umx_inference(om, wave, n_buf=N, want_taps=True)
T = N // 1024 + 1


def run(**kw):
    eng = pkg.Engine.from_file(path, N, **kw)
    if kw.get("tracks", 1) > 1:
        eng.infer_batch([wave] * kw["tracks"], pkg.FLAG_DEBUG_TAPS)
    else:
        eng.infer_segment(wave, pkg.FLAG_DEBUG_TAPS)
    r = {k: [eng.tap(k, t) for t in range(4)] for k in ("fc1", "lstm", "fc2", "mask")}
    r["x"] = eng.tap("x")[:, :2 * pkg.CROP].astype(np.float64)
    eng.close()
    return r


res = {"planes": run(gemm="planes"), "bf16x3": run(gemm="bf16x3"),
       "planes+batched LSTM": run(gemm="planes", tracks=2), "64 lanes (shipped)": run(gemm="planes", tracks=64)}
oracle = {"fc1": taps["fc1_out"], "lstm": taps["lstm_out"], "fc2": taps["fc2_out"]}


def g(t, n):
    return targets[t][n]["f32"].astype(np.float64)


def bn(y, t, name):
    return (y - g(t, name + ".running_mean")) / np.sqrt(g(t, name + ".running_var") + 1e-5) * g(t, name + ".weight") + g(t, name + ".bias")


def report(stage, name, got, want):
    e = got.astype(np.float64) - want
    print(f"{stage:6s} {name:30s} max abs {np.abs(e).max():10.3e}   rel L2 {np.linalg.norm(e) / np.linalg.norm(want):10.3e}")


for t in (0, 3):
    print(f"--- target {t}")
    for name, r in res.items():
        if "batched" in name:
            continue  # same GEMM kernels as "planes"
        a = r["x"] * np.tile(g(t, "input_scale"), 2) + np.tile(g(t, "input_mean"), 2)
        y = np.tanh(bn(a @ g(t, "fc1.weight").reshape(H, -1).T, t, "bn1"))
        report("fc1", name, r["fc1"][t], y)
    a = res["planes"]["x"] * np.tile(g(t, "input_scale"), 2) + np.tile(g(t, "input_mean"), 2)
    report("fc1", "CPU oracle (fp32)", oracle["fc1"][t], np.tanh(bn(a @ g(t, "fc1.weight").reshape(H, -1).T, t, "bn1")))
    for name, r in res.items():  # fc2 from the engine's own [fc1 | lstm]
        cat = np.concatenate([r["fc1"][t], r["lstm"][t]], axis=1).astype(np.float64)
        y = np.maximum(bn(cat @ g(t, "fc2.weight").reshape(H, -1).T, t, "bn2"), 0)
        report("fc2", name, r["fc2"][t], y)
    cat = np.concatenate([oracle["fc1"][t], oracle["lstm"][t]], axis=1).astype(np.float64)
    report("fc2", "CPU oracle (fp32)", oracle["fc2"][t], np.maximum(bn(cat @ g(t, "fc2.weight").reshape(H, -1).T, t, "bn2"), 0))
    # the recurrence in float64 from each engine's own fc1 output (zero initial state)
    lstm = torch.nn.LSTM(H, H // 2, num_layers=3, bidirectional=True).double()
    lstm.load_state_dict({f"{wn}_l{l}{sfx}": torch.from_numpy(g(t, f"lstm.{wn}_l{l}{sfx}").reshape(targets[t][f"lstm.{wn}_l{l}{sfx}"]["f32"].shape))
                          for l in range(3) for sfx in ("", "_reverse") for wn in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")})
    with torch.no_grad():
        for name in ("planes", "planes+batched LSTM", "64 lanes (shipped)"):
            want = lstm(torch.from_numpy(res[name]["fc1"][t].astype(np.float64))[:, None, :])[0][:, 0].numpy()
            report("lstm", name + (" (VALU kernel)" if name == "planes" else ""), res[name]["lstm"][t], want)
        want = lstm(torch.from_numpy(oracle["fc1"][t].astype(np.float64))[:, None, :])[0][:, 0].numpy()
        report("lstm", "CPU oracle (fp32)", oracle["lstm"][t], want)
print("end-to-end mask, planes vs bf16x3: max abs",
      max(float(np.abs(res['planes']['mask'][t] - res['bf16x3']['mask'][t]).max()) for t in range(4)))
