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


def main():
    pkg = ge.load_package()
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
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
