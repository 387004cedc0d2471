"""The reference's own dsp tests (test/test_dsp.cpp), restated against the oracle and the C++ host
wav reader.  Tolerance 1e-4 = NEAR_TOLERANCE (test_dsp.cpp:7)."""
from pathlib import Path

import numpy as np

from conftest import rel_l2

GOLD = Path(__file__).parent / "golden"
TOL = 1e-4


def test_load_mono_audio(pkg):  # test_dsp.cpp:10-23
    w, ch = pkg.wav_load(GOLD / "gspi_mono.wav")
    assert ch == 1 and w.shape == (2, 262144)
    assert w[0, 0] == w[1, 0] and w[0, 262143] == w[1, 262143]
    assert (w[0] == w[1]).all()  # mono duplicated, dsp.cpp:52-60


def test_load_stereo_audio(pkg):  # test_dsp.cpp:26-38
    w, ch = pkg.wav_load(GOLD / "gspi_stereo.wav")
    assert ch == 2 and w.shape == (2, 262144)
    assert w[0, 0] == w[1, 0] and w[0, 262143] == w[1, 262143]


def test_wav_write_read_roundtrip(pkg, tmp_path):  # dsp.cpp:80-101 float32 stereo out
    w, _ = pkg.wav_load(GOLD / "gspi_stereo.wav")
    pkg.wav_write(tmp_path / "o.wav", w[:, :5000])
    back, ch = pkg.wav_load(tmp_path / "o.wav")
    assert ch == 2 and (back == w[:, :5000]).all()


def test_wav_rejects_other_rate(pkg, tmp_path):  # dsp.cpp:27-33 (exit(1) there; an error code here)
    import struct
    raw = (GOLD / "gspi_mono.wav").read_bytes()
    i = raw.index(b"fmt ") + 12
    bad = raw[:i] + struct.pack("<I", 48000) + raw[i + 4:]
    (tmp_path / "48k.wav").write_bytes(bad)
    try:
        pkg.wav_load(tmp_path / "48k.wav")
    except pkg.HostError as e:
        assert e.code == 12 and "44100" in str(e)
    else:
        raise AssertionError("48 kHz file was accepted")


def test_stft_roundtrip_rand_waveform(po):  # test_dsp.cpp:41-80
    rng = np.random.default_rng(0)
    audio = rng.uniform(0, 1, (2, 4096)).astype(np.float32)
    spec = po.stft(audio)
    assert spec.shape == (2, 5, 2049)  # nb_bins 2049, nb_frames 4096/1024+1
    out = po.istft(spec, 4096)
    assert out.shape == audio.shape
    assert np.abs(out - audio).max() < TOL


def test_stft_roundtrip_glockenspiel(pkg, po):  # test_dsp.cpp:84-114
    audio, _ = pkg.wav_load(GOLD / "gspi_mono.wav")
    spec = po.stft(audio)
    assert spec.shape == (2, 257, 2049)
    out = po.istft(spec, audio.shape[1])
    assert np.abs(out - audio).max() < TOL


def _mag_phase_combine(po, audio):  # test_dsp.cpp:118-273
    spec = po.stft(audio)
    mag, ph = np.abs(spec), np.angle(spec)
    assert (mag >= 0).all()
    comb = (mag * np.cos(ph) + 1j * mag * np.sin(ph)).astype(np.complex64)  # polar_to_complex dsp.cpp:260-289
    assert np.abs(comb.real - spec.real).max() < TOL * max(1.0, np.abs(spec).max())
    assert np.abs(comb.imag - spec.imag).max() < TOL * max(1.0, np.abs(spec).max())
    out = po.istft(comb, audio.shape[1])
    assert np.abs(out - audio).max() < TOL


def test_magnitude_phase_combine_mono(pkg, po):
    audio, _ = pkg.wav_load(GOLD / "gspi_mono.wav")
    _mag_phase_combine(po, audio)


def test_magnitude_phase_combine_stereo(pkg, po):
    audio, _ = pkg.wav_load(GOLD / "gspi_stereo.wav")
    _mag_phase_combine(po, audio)


def test_window_and_sumsq(po):  # dsp.hpp:61-101
    w = po.hann_window()
    ref = 0.5 * (1 - np.cos(2 * np.pi * np.arange(4096) / 4096))
    assert np.abs(w - ref).max() < 1e-6 and w[0] == 0.0
    nw = po.window_sumsq(5)
    assert nw.shape == (4096 + 4 * 1024,)
    assert abs(nw[2048 + 1024] - 1.5) < 1e-5  # 4 overlapping hann^2 at 75 % overlap sum to 1.5


def test_short_chunk_is_zero_padded_to_full_buffer(po):  # dsp.cpp:214-217, SURVEY a3
    rng = np.random.default_rng(1)
    n, n_buf = 5000, 16384
    w = rng.uniform(-1, 1, (2, n)).astype(np.float32)
    full = np.zeros((2, n_buf), np.float32)
    full[:, :n] = w
    a = po.stft(w, n_buf)
    b = po.stft(full, n_buf)
    assert a.shape == (2, 17, 2049)
    assert rel_l2(a, b) < 1e-7
    out = po.istft(a, n, n_buf)
    assert out.shape == (2, n) and np.abs(out - w).max() < TOL
