"""Generates the golden fixtures under tests/golden/ (run in the build container; committed).

The reference cannot be built or imported here (Eigen / libnyquist / openunmix are absent:
SURVEY.md F1), so these vectors come from INDEPENDENT implementations of the same math available
in this container -- they pin the oracle (oracle/umx_oracle.cpp) against restatement errors:
  stft_f64.npz     numpy float64 STFT (np.pad mode='symmetric', F7) + iSTFT round trip
  stft_probe.npz   the reference's own known-answer probe (scripts/compare-torch-stft.py:9-12:
                   2x4096 zeros, samples 0..19 = +-0.5) -- centre frame 2 via torch.stft
  lstm_torch.npz   torch.nn.LSTM(H, H/2, 3 layers, bidirectional) incl. carried (h, c) state (F3)
  dense_torch.npz  torch Linear(bias=False) + BatchNorm1d.eval() stack of inference.cpp:75-185
  wiener_f64.npz   numpy float64 restatement of wiener.cpp:92-425 incl. F5 / F6, subset of bins
  target_network_f64.npz  the WHOLE per-target network of inference.cpp:75-185 in torch float64: x*scale+mean (F8) ->
                   fc1/bn1/tanh -> 3-layer BiLSTM from a non-zero carried state -> [fc1|lstm] -> fc2/bn2/relu -> fc3/bn3
                   -> *output_scale+output_mean, relu -> mask x mix_mag; every stage is stored
  split_f64.npz    numpy float64 restatement of split_inference's chunking, triangular transition weights, weighted
                   overlap-add and normalisation (umx.cpp:181-273, sum_weight zeroed = F4 fixed) and of
                   shift_inference's padding/crop (umx.cpp:115-147) around a known per-segment function
  quantizer_ref.npz  (round 4) THE REFERENCE'S OWN quantiser: `quantize` / `dequantize` are lifted out of
                   /root/reference/scripts/convert-umx-pth-to-ggml.py:13-34 by AST at generation time (the script's top-level
                   `import openunmix` is the only thing that keeps it from being imported here; the two functions are pure
                   numpy) and RUN on seeded tensors: every tensor of one synthetic target at hidden 16 with the script's own
                   u8 / u16 rule (:146-150), plus constant, single-element and extreme-range tensors.  Stored: (scale, offset)
                   and the sha256 of q and of the dequantised fp32 bytes per tensor, the small tensors in full.  No reference
                   source text is stored -- only what the functions returned.
Only seeds + expected outputs are stored; inputs are regenerated from the seed by the tests.
gspi_mono.wav / gspi_stereo.wav are the reference's own test data files (test/data/), copied as data.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

HERE = Path(__file__).resolve().parent
NB, NFFT, HOP = 2049, 4096, 1024


def hann():
    return 0.5 * (1 - np.cos(2 * np.pi * np.arange(NFFT) / NFFT))


def stft_f64(wave, n_buf=None):
    """float64 STFT with the reference's buffer semantics (dsp.cpp:109-176)."""
    n = wave.shape[1]
    n_buf = n if n_buf is None else n_buf
    T = n_buf // HOP + 1
    out = np.zeros((2, T, NB), np.complex128)
    w = hann()
    for c in range(2):
        buf = np.zeros(n_buf + NFFT)
        buf[2048:2048 + n] = wave[c]
        buf[:2048] = buf[2048:4096][::-1].copy()
        buf[-2048:] = buf[-4096:-2048][::-1].copy()
        for f in range(T):
            out[c, f] = np.fft.rfft(buf[f * HOP:f * HOP + NFFT] * w)
    return out


def istft_f64(spec, n, n_buf=None):
    n_buf = n if n_buf is None else n_buf
    T = n_buf // HOP + 1
    w = hann()
    nw = np.zeros(NFFT + HOP * (T - 1))
    for f in range(T):
        nw[f * HOP:f * HOP + NFFT] += w * w
    out = np.zeros((2, n))
    for c in range(2):
        buf = np.zeros(n_buf + NFFT)
        for f in range(T):
            s = spec[c, f].copy()
            s[0] = s[0].real
            s[-1] = s[-1].real
            fr = np.fft.irfft(s, NFFT) * NFFT  # unscaled inverse
            buf[f * HOP:f * HOP + NFFT] += fr * w / NFFT / (nw[f * HOP:f * HOP + NFFT] + 1e-8)
        out[c] = buf[2048:2048 + n]
    return out


def wiener_f64(X, mags, eps=1e-10, scale=10.0):
    """float64 restatement of wiener.cpp:92-425 (one EM iteration) with F5 and F6."""
    T = X.shape[1]
    ph = np.angle(X)
    y = [m * np.exp(1j * ph) for m in mags]
    max_abs = max(1.0, np.abs(X).max() / scale)
    X = X / max_abs
    y = [yy / max_abs for yy in y]
    v = [0.5 * ((yy.real + yy.imag) ** 2).sum(axis=0) for yy in y]  # (T,B)  F5
    R = []
    for j in range(4):
        w = eps + v[j].sum(axis=0)  # (B,)
        Rj = np.einsum("afb,cfb->bac", y[j], np.conj(y[j])) / w[:, None, None]
        R.append(Rj)
    reg = np.sqrt(eps) * np.eye(2)
    Cxx = sum(reg[None, None] + v[j][:, :, None, None] * R[j][None] for j in range(4))  # F6: 4x reg
    inv = np.linalg.inv(Cxx)
    out = []
    for j in range(4):
        G = v[j][:, :, None, None] * np.einsum("bac,fbcd->fbad", R[j], inv)
        yj = np.einsum("fbac,cfb->afb", G, X)
        out.append(yj * max_abs)
    return out


def target_network_f64(wt, H, x, mix_mag, state):
    """inference.cpp:75-185 for one target in torch float64.  wt: that target's tensors (fp32 arrays), x (T,2974),
    mix_mag (2,T,2049), state [3][2][2][H/2] (h, c per layer and direction) -> dict of stages + the new state."""
    f = lambda a: torch.from_numpy(np.asarray(a, np.float64))  # noqa: E731
    with torch.no_grad():
        xs = f(x) * f(np.tile(wt["input_scale"], 2)) + f(np.tile(wt["input_mean"], 2))  # inference.cpp:78-83 (F8)

        def bn(y, name):  # inference.cpp:93-97: ((y - mean) / sqrt(var + 1e-5)) * weight + bias
            return (y - f(wt[name + ".running_mean"])) / torch.sqrt(f(wt[name + ".running_var"]) + 1e-5) * f(wt[name + ".weight"]) \
                + f(wt[name + ".bias"])
        a1 = torch.tanh(bn(xs @ f(wt["fc1.weight"]).T, "bn1"))
        lstm = torch.nn.LSTM(H, H // 2, num_layers=3, bidirectional=True).double()
        lstm.load_state_dict({f"{wn}_l{l}{sfx}": f(wt[f"lstm.{wn}_l{l}{sfx}"]) for l in range(3) for sfx in ("", "_reverse")
                              for wn in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")})
        st = f(state).reshape(3, 2, 2, H // 2)
        h0, c0 = st[:, :, 0].reshape(6, 1, H // 2).contiguous(), st[:, :, 1].reshape(6, 1, H // 2).contiguous()
        lo, (hn, cn) = lstm(a1[:, None, :], (h0, c0))  # lstm.cpp:101-179, state index layer*2+dir, gates i|f|g|o
        lo = lo[:, 0]
        a2 = torch.relu(bn(torch.cat([a1, lo], dim=1) @ f(wt["fc2.weight"]).T, "bn2"))  # inference.cpp:118-140
        a3 = bn(a2 @ f(wt["fc3.weight"]).T, "bn3")  # inference.cpp:143-156
        mask = torch.relu(a3 * f(np.tile(wt["output_scale"], 2)) + f(np.tile(wt["output_mean"], 2)))  # :161-166
        mm = f(mix_mag)
        tm = torch.stack([mask[:, :NB] * mm[0], mask[:, NB:] * mm[1]])  # inference.cpp:173-183
        new_state = torch.stack([hn[:, 0].reshape(3, 2, H // 2), cn[:, 0].reshape(3, 2, H // 2)], dim=2)
    return {"fc1": a1.numpy(), "lstm": lo.numpy(), "fc2": a2.numpy(), "mask": mask.numpy(), "target_mag": tm.numpy(),
            "state": new_state.numpy()}


def transition_weight_f64(N):
    """umx.cpp:197-206: 1..N/2 then N/2..1, divided by the maximum, ^TRANSITION_POWER (= 1)."""
    half = N // 2
    w = np.concatenate([np.arange(1, half + 1), np.arange(N - half, 0, -1)]).astype(np.float64)
    return w / w.max()


def split_f64(audio, N, segment_fn):
    """umx.cpp:152-295 (sum_weight zero-initialised, F4): audio (2,L) float64; segment_fn(chunk (2,n), index) -> 4 x (2,n)."""
    L = audio.shape[1]
    stride = int((1 - 0.25) * N)  # umx.cpp:181
    w = transition_weight_f64(N)
    out = np.zeros((4, 2, L))
    sw = np.zeros(L)
    offs = []
    for i, off in enumerate(range(0, L, stride)):
        n = min(N, L - off)  # umx.cpp:214-217
        stems = segment_fn(audio[:, off:off + n], i)
        for t in range(4):
            out[t, :, off:off + n] += w[:n] * stems[t]  # umx.cpp:243-253 (weight index = k % chunk_length, k < n)
        sw[off:off + n] += w[:n]
        offs.append(off)
    return out / sw, sw, offs  # umx.cpp:264-273


def shift_f64(audio, N, offset, segment_fn, max_shift=22050):
    """umx.cpp:99-150: delay by `offset` inside a zero buffer of length L + max_shift - offset, split, crop.  (The
    reference's buffer is too short for its own block write when offset > max_shift / 2; sized to hold it.)"""
    L = audio.shape[1]
    buf = np.zeros((2, L + max(max_shift - offset, offset)))
    buf[:, offset:offset + L] = audio
    out, _, _ = split_f64(buf, N, segment_fn)
    return out[:, :, offset:offset + L]


def pseudo_segment(chunk, i):
    """A known stand-in for umx_inference in the driver goldens: stem t = (t+1) * chunk + 0.001 * (i+1) * ramp."""
    n = chunk.shape[1]
    ramp = np.linspace(-1.0, 1.0, n) if n > 1 else np.zeros(1)
    return [(t + 1) * chunk + 0.001 * (i + 1) * ramp for t in range(4)]


def new_goldens(pkg):
    # ---- 6. the whole target network in float64, from a non-zero carried state
    H, T, tg = 64, 6, 3
    W = pkg.ggml.synth_weights(H, seed=606)
    rng = np.random.default_rng(607)
    x = (np.abs(rng.standard_normal((T, 2974))) * 20).astype(np.float32)
    mix = (np.abs(rng.standard_normal((2, T, NB))) * 30).astype(np.float32)
    state = (rng.standard_normal(12 * (H // 2)) * 0.3).astype(np.float32)
    g = target_network_f64(W[tg], H, x, mix, state)
    np.savez_compressed(HERE / "target_network_f64.npz", hidden=H, T=T, target=tg, wseed=606, xseed=607,
                        fc1=g["fc1"].astype(np.float32), lstm=g["lstm"].astype(np.float32), fc2=g["fc2"].astype(np.float32),
                        mask=g["mask"].astype(np.float32), target_mag=g["target_mag"].astype(np.float32),
                        state=g["state"].astype(np.float32).ravel())
    # ---- 7. segment drivers in float64 around a known per-segment function
    N = 4096
    rng = np.random.default_rng(708)
    L = int(N * 3.4)
    audio = rng.uniform(-1, 1, (2, L)).astype(np.float32)
    out, sw, offs = split_f64(audio.astype(np.float64), N, pseudo_segment)
    # shift_inference with the reference's constants: its unseeded offset 4033, and one beyond max_shift / 2 (where
    # the reference overruns its buffer; see shift_f64)
    sh = shift_f64(audio.astype(np.float64)[:, :N // 3], N, 4033, pseudo_segment)
    sh2 = shift_f64(audio.astype(np.float64)[:, :N // 3], N, 20000, pseudo_segment)
    np.savez_compressed(HERE / "split_f64.npz", seed=708, N=N, L=L, weight=transition_weight_f64(N).astype(np.float32),
                        sum_weight=sw[::5].astype(np.float32), offsets=np.array(offs), every=5, out=out[:, :, ::5].astype(np.float32),
                        shift_len=N // 3, shift_out_4033=sh[:, :, ::3].astype(np.float32), shift_out_20000=sh2[:, :, ::3].astype(np.float32))
    print("new golden fixtures written")


QUANT_HIDDEN, QUANT_SEED = 16, 909


def quant_special_cases():
    """Inputs of the special cases, regenerated by the test from this function (seeded / closed form)."""
    rng = np.random.default_rng(910)
    return {
        "single": np.array([0.37], np.float32),
        "two_equal_steps": np.array([-1.0, 1.0], np.float32),
        "constant": np.full(7, 0.25, np.float32),  # scale = 0: 0/0 in the reference quantiser (NaN -> integer cast), whatever it gives
        "tiny_range": (1.0 + rng.uniform(0, 1e-6, 33)).astype(np.float32),
        "huge_range": (rng.standard_normal(65) * 1e30).astype(np.float32),
        "denormal": (rng.uniform(-1, 1, 40) * 1e-41).astype(np.float32),
        "ties": (np.arange(0, 510, dtype=np.float32) * 0.5),  # x.5 quotients: numpy rounds half to even
        "negative_only": (-np.abs(rng.standard_normal(50)) - 3).astype(np.float32),
    }


def reference_quantiser():
    """`quantize`, `dequantize` of the reference's converter, lifted by AST (build container only)."""
    import ast
    src = Path("/root/reference/scripts/convert-umx-pth-to-ggml.py").read_text()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("quantize", "dequantize")]
    assert [f.name for f in fns] == ["quantize", "dequantize"]
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "convert-umx-pth-to-ggml.py", "exec"), ns)  # noqa: S102
    return ns["quantize"], ns["dequantize"]


def quant_goldens(pkg):
    import hashlib
    import warnings
    rq, rd = reference_quantiser()
    W = pkg.ggml.synth_weights(QUANT_HIDDEN, seed=QUANT_SEED)[0]
    out = {"hidden": QUANT_HIDDEN, "wseed": QUANT_SEED, "numpy": np.__version__}
    names = pkg.ggml.tensor_names()
    scales, offsets, qsha, dsha = [], [], [], []
    for nm in names:
        data = W[nm].astype(np.float32)
        # the converter's own rule, convert-umx-pth-to-ggml.py:146-150
        if any([x in nm for x in ["bn2", "bn3", "fc2", "fc3"]]):
            q, scale, offset = rq(data, qtype=np.uint16)
        else:
            q, scale, offset = rq(data)
        import struct
        scale32, offset32 = struct.unpack("ff", struct.pack("ff", scale, offset))  # what :154 writes to the file
        deq = np.asarray(rd(q, np.float32(scale32), np.float32(offset32)))
        assert deq.dtype == np.float32, deq.dtype
        scales.append(scale32)
        offsets.append(offset32)
        qsha.append(hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest())
        dsha.append(hashlib.sha256(np.ascontiguousarray(deq).tobytes()).hexdigest())
        if data.ndim == 1:
            out["q/" + nm] = q
            out["deq/" + nm] = deq
    out.update(names=np.array(names), scale=np.array(scales, np.float32), offset=np.array(offsets, np.float32),
               q_sha256=np.array(qsha), deq_sha256=np.array(dsha))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for nm, a in quant_special_cases().items():
            for qt in (np.uint8, np.uint16):
                q, scale, offset = rq(a, qtype=qt)
                key = f"special/{nm}/{np.dtype(qt).name}"
                out[key + "/q"] = q
                out[key + "/scale"] = np.float32(scale)
                out[key + "/offset"] = np.float32(offset)
                out[key + "/deq"] = np.asarray(rd(q, np.float32(scale), np.float32(offset)), np.float32)
    np.savez_compressed(HERE / "quantizer_ref.npz", **out)
    print("quantizer_ref.npz written:", len(names), "tensors +", len(quant_special_cases()) * 2, "special cases")


def main():
    pkg = ge.load_package()
    if len(sys.argv) > 1 and sys.argv[1] == "new":  # only the fixtures added in round 2
        return new_goldens(pkg)
    if len(sys.argv) > 1 and sys.argv[1] == "quant":  # round 4: the reference's own quantiser, run here
        return quant_goldens(pkg)
    # ---- 1. float64 STFT / iSTFT
    rng = np.random.default_rng(101)
    n, n_buf = 6000, 8192
    wave = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    S = stft_f64(wave.astype(np.float64), n_buf)
    back = istft_f64(S, n, n_buf)
    np.savez_compressed(HERE / "stft_f64.npz", seed=101, n=n, n_buf=n_buf, spec=S.astype(np.complex64),
                        roundtrip_err=np.abs(back - wave).max())
    # ---- 2. the reference's torch-stft probe (compare-torch-stft.py:9-23)
    a = torch.zeros((2, 4096))
    for i in range(20):
        a[:, i] = 0.5 if i % 2 == 0 else -0.5
    win = torch.hann_window(4096, periodic=True)
    st = torch.stft(a, n_fft=4096, hop_length=1024, window=win, center=True, pad_mode="reflect",
                    return_complex=True, normalized=False, onesided=True)  # (2, 2049, 5)
    np.savez_compressed(HERE / "stft_probe.npz", centre_frame=st[:, :, 2].numpy().astype(np.complex64))
    # ---- 3. LSTM vs torch
    H, T = 64, 24
    W = pkg.ggml.synth_weights(H, seed=202)
    tg = 1
    lstm = torch.nn.LSTM(H, H // 2, num_layers=3, bidirectional=True)
    sd = {}
    for l in range(3):
        for sfx in ("", "_reverse"):
            for wn in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                sd[f"{wn}_l{l}{sfx}"] = torch.from_numpy(W[tg][f"lstm.{wn}_l{l}{sfx}"])
    lstm.load_state_dict(sd)
    x = np.random.default_rng(203).standard_normal((T, H)).astype(np.float32)
    with torch.no_grad():
        o1, (h1, c1) = lstm(torch.from_numpy(x)[:, None, :])
        o2, (h2, c2) = lstm(torch.from_numpy(x[::-1].copy())[:, None, :], (h1, c1))
    np.savez_compressed(HERE / "lstm_torch.npz", hidden=H, T=T, wseed=202, xseed=203, target=tg,
                        out1=o1[:, 0].numpy(), out2=o2[:, 0].numpy(),
                        h2=h2[:, 0].numpy(), c2=c2[:, 0].numpy())
    # ---- 4. dense stack vs torch (lstm output replaced by a seeded tensor to isolate the stack)
    T = 10
    xin = np.abs(np.random.default_rng(204).standard_normal((T, 2974))).astype(np.float32) * 20
    tg = 2
    wt = W[tg]
    with torch.no_grad():
        xt = torch.from_numpy(xin)
        sc = torch.from_numpy(np.tile(wt["input_scale"], 2))
        mn = torch.from_numpy(np.tile(wt["input_mean"], 2))
        xs = xt * sc + mn  # F8 order
        fc1 = torch.nn.Linear(2974, H, bias=False)
        fc1.weight.copy_(torch.from_numpy(wt["fc1.weight"]))
        bn1 = torch.nn.BatchNorm1d(H).eval()
        bn1.weight.copy_(torch.from_numpy(wt["bn1.weight"]))
        bn1.bias.copy_(torch.from_numpy(wt["bn1.bias"]))
        bn1.running_mean.copy_(torch.from_numpy(wt["bn1.running_mean"]))
        bn1.running_var.copy_(torch.from_numpy(wt["bn1.running_var"]))
        a1 = torch.tanh(bn1(fc1(xs)))
    np.savez_compressed(HERE / "dense_torch.npz", hidden=H, T=T, wseed=202, xseed=204, target=tg, fc1_out=a1.numpy())
    # ---- 5. Wiener float64, T crosses the 200-frame batch boundary; keep a subset of bins
    T = 230
    rng = np.random.default_rng(305)
    X = (rng.standard_normal((2, T, NB)) + 1j * rng.standard_normal((2, T, NB))).astype(np.complex64) * 30
    mags = [(rng.uniform(0, 1.5, (2, T, NB)) * np.abs(X)).astype(np.float32) for _ in range(4)]
    yy = wiener_f64(X.astype(np.complex128), [m.astype(np.float64) for m in mags])
    bins = np.array([0, 1, 2, 7, 100, 511, 1024, 1486, 1487, 2000, 2047, 2048])
    np.savez_compressed(HERE / "wiener_f64.npz", seed=305, T=T, bins=bins,
                        y=np.stack([y[:, :, bins] for y in yy]).astype(np.complex64))
    new_goldens(pkg)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
