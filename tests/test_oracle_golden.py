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


def test_whole_target_network_vs_float64_torch(pkg, po):
    """inference.cpp:75-185 end to end for one target -- input scale (F8 order), fc1/bn1/tanh, the 3-layer BiLSTM from a
    NON-ZERO carried state, skip concat, fc2/bn2/relu, fc3/bn3, output scale + relu, mask x mix magnitude -- against an
    independent torch float64 evaluation (tests/golden/make_golden.py::target_network_f64).  Every stage is pinned,
    including fc2 / fc3 / bn3 / output scaling, which the earlier goldens did not reach."""
    g = np.load(GOLD / "target_network_f64.npz")
    H, T, tg = int(g["hidden"]), int(g["T"]), int(g["target"])
    m = po.Model.from_arrays(H, _weights(pkg, int(g["wseed"]), H))
    rng = np.random.default_rng(int(g["xseed"]))
    x = (np.abs(rng.standard_normal((T, 2974))) * 20).astype(np.float32)
    mix = (np.abs(rng.standard_normal((2, T, 2049))) * 30).astype(np.float32)
    state = (rng.standard_normal(12 * (H // 2)) * 0.3).astype(np.float32)
    got = po.target_network(m, tg, x, mix, state)
    assert np.abs(got["fc1"] - g["fc1"]).max() < 5e-6
    assert np.abs(got["lstm"] - g["lstm"]).max() < 5e-6
    assert np.abs(state - g["state"]).max() < 5e-6  # the carried (h, c) after the segment
    for k in ("fc2", "mask", "target_mag"):
        assert rel_l2(got[k], g[k]) < 2e-6, k
    assert (got["mask"] >= 0).all() and (got["mask"] == 0).any() and (got["mask"] > 0).any()  # the relu is exercised


def _pseudo_segment(chunk, i):
    """the known per-segment function of the driver goldens (make_golden.py::pseudo_segment), in float32"""
    n = chunk.shape[1]
    ramp = np.linspace(-1.0, 1.0, n) if n > 1 else np.zeros(1)
    return [((t + 1) * chunk.astype(np.float64) + 0.001 * (i + 1) * ramp).astype(np.float32) for t in range(4)]


def test_segment_drivers_vs_float64_numpy(pkg, po):
    """split_inference's chunking, triangular weights, weighted overlap-add and normalisation (umx.cpp:181-273) and
    shift_inference's delay / crop (umx.cpp:115-147) against a numpy float64 restatement around a known per-segment
    function: pins the C++17 host drivers (host/split.cpp) directly, and the oracle's drivers through its own
    per-segment outputs."""
    g = np.load(GOLD / "split_f64.npz")
    N, L, ev = int(g["N"]), int(g["L"]), int(g["every"])
    audio = np.random.default_rng(int(g["seed"])).uniform(-1, 1, (2, L)).astype(np.float32)
    # transition weights and the segment plan
    hl = pkg.host_lib()
    w = np.array([hl.umx_transition_weight(k, N, N) for k in range(N)], np.float32)
    assert np.abs(w - g["weight"]).max() < 1e-7
    offs, lens = pkg.segment_plan(L, N)
    assert offs == list(g["offsets"]) and lens[-1] == L - offs[-1] and all(n == N for n in lens[:-1])
    # host driver with the known per-segment function
    calls = []

    def seg(chunk):
        calls.append(chunk.shape[1])
        return _pseudo_segment(chunk, len(calls) - 1)
    be = pkg.make_backend(seg)
    out = pkg.split_inference(be, audio, N)
    assert calls == lens
    for t in range(4):
        assert np.abs(out[t][:, ::ev] - g["out"][t]).max() < 2e-6, t
    # shift_inference: the reference's own offset, and one beyond max_shift / 2 (a buffer overrun in the reference)
    n3 = int(g["shift_len"])
    for off in (4033, 20000):
        calls.clear()
        sh = pkg.shift_inference(be, audio[:, :n3], N, offset=off)
        for t in range(4):
            assert np.abs(sh[t][:, ::3] - g[f"shift_out_{off}"][t]).max() < 2e-6, (off, t)
    # the oracle's split driver: its own per-segment outputs (carried state) blended with the golden weights in float64
    H = 64
    m = po.Model.from_arrays(H, _weights(pkg, 7, H))
    st = po.stream_state(H)
    acc, sw = np.zeros((4, 2, L)), np.zeros(L)
    wt = g["weight"].astype(np.float64)
    for off, n in zip(offs, lens):
        stems, _ = po.umx_inference(m, audio[:, off:off + n], n_buf=N, state=st)
        for t in range(4):
            acc[t, :, off:off + n] += wt[:n] * stems[t]
        sw[off:off + n] += wt[:n]
    ref = acc / sw
    got = po.split_inference(m, audio, N)
    for t in range(4):
        assert np.abs(got[t] - ref[t]).max() < 2e-6, t
    assert np.abs(sw[::ev] - g["sum_weight"]).max() < 1e-6


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """`make -C oracle sanitize`: the restatement's entry points on a short segment / ragged track under ASan + UBSan
    (the reference has no sanitizer target either; the checker should not be the thing with the stray read)."""
    import shutil
    import subprocess
    from pathlib import Path
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no compiler")
    p = subprocess.run(["make", "-C", str(Path(__file__).resolve().parent.parent / "oracle"), "sanitize"],
                       capture_output=True, text=True, timeout=600)
    if "cannot find -lasan" in p.stderr or "cannot find -lubsan" in p.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert p.returncode == 0 and "selftest: ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
