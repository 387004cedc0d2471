"""Pins the oracle against the committed golden vectors (tests/golden/make_golden.py): float64
numpy, torch.nn.LSTM / Linear / BatchNorm1d, and the reference's own torch-stft probe."""
from pathlib import Path

import numpy as np

from conftest import rel_l2

GOLD = Path(__file__).parent / "golden"


def test_stft_vs_float64_numpy(po):
    g = np.load(GOLD / "stft_f64.npz")
    n, n_buf = int(g["n"]), int(g["n_buf"])
    wave = np.random.default_rng(int(g["seed"])).uniform(-1, 1, (2, n)).astype(np.float32)
    S = po.stft(wave, n_buf)
    assert rel_l2(S, g["spec"]) < 1e-6  # fp32 FFT vs float64 (<= 1e-5 required)
    assert np.abs(po.istft(S, n, n_buf) - wave).max() < 1e-5


def test_stft_probe_centre_frame(po):
    """scripts/compare-torch-stft.py:9-23: 2x4096 zeros with samples 0..19 = +-0.5; the centre
    frame (2 of 5) touches no padding, so torch.stft(reflect) is a valid golden despite F7."""
    a = np.zeros((2, 4096), np.float32)
    a[:, :20] = np.where(np.arange(20) % 2 == 0, 0.5, -0.5)
    S = po.stft(a)
    g = np.load(GOLD / "stft_probe.npz")["centre_frame"]  # (2, 2049)
    assert np.abs(S[:, 2, :] - g).max() < 1e-5
    assert np.abs(S[0] - S[1]).max() == 0.0


def _weights(pkg, seed, hidden):
    return pkg.ggml.synth_weights(hidden, seed=seed)


def test_lstm_vs_torch_with_state_carry(pkg, po):
    g = np.load(GOLD / "lstm_torch.npz")
    H, T, tg = int(g["hidden"]), int(g["T"]), int(g["target"])
    m = po.Model.from_arrays(H, _weights(pkg, int(g["wseed"]), H))
    x = np.random.default_rng(int(g["xseed"])).standard_normal((T, H)).astype(np.float32)
    st = np.zeros(12 * (H // 2), np.float32)
    o1 = po.lstm_forward(m, tg, x, st)
    assert np.abs(o1 - g["out1"]).max() < 2e-6
    # second call continues from the carried state (F3), including the backward-chain quirk
    o2 = po.lstm_forward(m, tg, x[::-1].copy(), st)
    assert np.abs(o2 - g["out2"]).max() < 2e-6
    s = st.reshape(3, 2, 2, H // 2)
    assert np.abs(s[:, :, 0].reshape(6, -1) - g["h2"]).max() < 2e-6
    assert np.abs(s[:, :, 1].reshape(6, -1) - g["c2"]).max() < 2e-6


def test_dense_fc1_bn_tanh_vs_torch(pkg, po):
    g = np.load(GOLD / "dense_torch.npz")
    H, T, tg = int(g["hidden"]), int(g["T"]), int(g["target"])
    m = po.Model.from_arrays(H, _weights(pkg, int(g["wseed"]), H))
    x = np.abs(np.random.default_rng(int(g["xseed"])).standard_normal((T, 2974))).astype(np.float32) * 20
    import ctypes as C
    fc1 = np.empty((T, H), np.float32)
    mix = np.zeros((2 * T * 2049,), np.float32)
    st = np.zeros(12 * (H // 2), np.float32)
    fp = C.POINTER(C.c_float)
    po.lib().oracle_target_network(m.h, tg, x.ctypes.data_as(fp), mix.ctypes.data_as(fp), T,
                                   st.ctypes.data_as(fp), fc1.ctypes.data_as(fp), None, None, None)
    assert np.abs(fc1 - g["fc1_out"]).max() < 5e-6


def test_wiener_vs_float64(po):
    g = np.load(GOLD / "wiener_f64.npz")
    T, bins = int(g["T"]), g["bins"]
    rng = np.random.default_rng(int(g["seed"]))
    X = (rng.standard_normal((2, T, 2049)) + 1j * rng.standard_normal((2, T, 2049))).astype(np.complex64) * 30
    mags = [(rng.uniform(0, 1.5, (2, T, 2049)) * np.abs(X)).astype(np.float32) for _ in range(4)]
    y = po.wiener(X, mags)
    for j in range(4):
        assert rel_l2(y[j][:, :, bins], g["y"][j]) < 2e-5


def test_wiener_quirks_are_load_bearing(po):
    """F5: v uses (Re+Im)^2.  A y rotated by 90 degrees has the same |y|^2 but a different v, so the
    output must change; with the textbook PSD it would only rotate."""
    rng = np.random.default_rng(5)
    T = 8
    X = (rng.standard_normal((2, T, 2049)) + 1j * rng.standard_normal((2, T, 2049))).astype(np.complex64)
    mags = [(rng.uniform(0.1, 1, (2, T, 2049))).astype(np.float32) for _ in range(4)]
    y1 = po.wiener(X, mags)
    y2 = po.wiener((X * 1j).astype(np.complex64), mags)
    assert rel_l2(y2[0], y1[0] * 1j) > 1e-2


def test_inference_pipeline_shapes_and_state(pkg, po):
    H, n_buf = 64, 8 * 1024
    m = po.Model.from_arrays(H, _weights(pkg, 7, H))
    wave = pkg.ggml.synth_audio(5000, 2)
    st = po.stream_state(H)
    outs, taps = po.umx_inference(m, wave, n_buf=n_buf, state=st, want_taps=True)
    assert all(o.shape == (2, 5000) for o in outs)
    assert taps["spec"].shape == (2, 9, 2049) and taps["x"].shape == (9, 2974)
    assert np.abs(st).max() > 0  # the stream state moved (F3)
    # x is |spec| cropped to 1487 bins per channel and stacked L|R (inference.cpp:58-68)
    assert np.allclose(taps["x"][:, :1487], np.abs(taps["spec"][0, :, :1487]), atol=1e-5)
    assert np.allclose(taps["x"][:, 1487:], np.abs(taps["spec"][1, :, :1487]), atol=1e-5)
    # mask * mix_mag (inference.cpp:175-183)
    for t in range(4):
        tm = taps["target_mag"][t]
        assert np.allclose(tm[0], taps["mask"][t][:, :2049] * taps["mix_mag"][0], atol=1e-4)
        assert np.allclose(tm[1], taps["mask"][t][:, 2049:] * taps["mix_mag"][1], atol=1e-4)
    # a second segment gives a different result because of the carried state
    outs2, _ = po.umx_inference(m, wave, n_buf=n_buf, state=st)
    assert np.abs(outs2[0] - outs[0]).max() > 1e-6
