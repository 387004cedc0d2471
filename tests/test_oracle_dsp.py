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


def _wav_bytes(samples, channels, bits, fmt_tag, extensible=False, extra_chunk=False):
    """A RIFF/WAVE file built by hand: samples (n, channels) ints (PCM) or floats (tag 3), little endian."""
    import struct
    bps = bits // 8
    if fmt_tag == 3:
        body = np.asarray(samples, "<f4").tobytes()
    elif bits == 24:
        a = np.asarray(samples, np.int64).ravel()
        body = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in a)
    else:
        body = np.asarray(samples, {16: "<i2", 32: "<i4"}[bits]).tobytes()
    if extensible:  # WAVE_FORMAT_EXTENSIBLE: tag 0xFFFE, the real tag is the first word of the sub-format GUID
        fmt = struct.pack("<HHIIHH", 0xFFFE, channels, 44100, 44100 * bps * channels, bps * channels, bits)
        fmt += struct.pack("<HHI", 22, bits, 3 if channels == 2 else 4)
        fmt += struct.pack("<H", fmt_tag) + bytes.fromhex("000000001000800000AA00389B71")
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, 44100, 44100 * bps * channels, bps * channels, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunk:  # an odd-sized chunk the reader must skip (with its pad byte)
        chunks += b"LIST" + struct.pack("<I", 5) + b"hello" + b"\0"
    chunks += b"data" + struct.pack("<I", len(body)) + body
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_wav_reader_encodings(pkg, tmp_path):
    """libnyquist (dsp.cpp:23-25) decodes more than 16-bit PCM.  PCM 16 / 24 / 32, 32-bit float, and the
    same inside WAVE_FORMAT_EXTENSIBLE, mono and stereo, built by hand here: every decoder must return the same
    waveform within its own quantisation step; integer full scale maps to 1.0 (16-bit: / 32767, the value the shipped
    glockenspiel files decode to under the reference's first-and-last-sample test; 24 / 32-bit: / 2^23, / 2^31)."""
    rng = np.random.default_rng(12)
    n = 3000
    x = rng.uniform(-0.9, 0.9, (n, 2))
    cases = []
    for ch in (1, 2):
        xs = x[:, :ch]
        cases += [(np.round(xs * 32767), ch, 16, 1, False, 1 / 32767, 1 / 32767),
                  (np.round(xs * 8388607), ch, 24, 1, False, 1 / 8388608, 1 / 8388608),
                  (np.round(xs * 2147483647), ch, 32, 1, False, 1 / 2147483648, 1e-7),
                  (xs, ch, 32, 3, False, 1.0, 1e-7),
                  (np.round(xs * 8388607), ch, 24, 1, True, 1 / 8388608, 1 / 8388608),
                  (xs, ch, 32, 3, True, 1.0, 1e-7)]
    for i, (smp, ch, bits, tag, ext, scale, tol) in enumerate(cases):
        f = tmp_path / f"c{i}.wav"
        f.write_bytes(_wav_bytes(smp, ch, bits, tag, ext, extra_chunk=(i % 2 == 1)))
        w, nch = pkg.wav_load(f)
        assert nch == ch and w.shape == (2, n), (i, w.shape)
        want = x[:, :ch] if tag == 3 else smp * scale
        assert np.abs(w[0] - want[:, 0]).max() <= tol * 1.01, (i, bits, tag, ext)
        assert np.abs(w[1] - want[:, ch - 1]).max() <= tol * 1.01  # mono is duplicated (dsp.cpp:52-60)
        assert np.abs(w[0] - x[:, 0]).max() < 2 * max(tol, 1e-7) + 1e-7
    # extremes of every integer format
    for bits, lo, hi, div in ((16, -32768, 32767, 32767.0), (24, -8388608, 8388607, 8388608.0), (32, -2147483648, 2147483647, 2147483648.0)):
        f = tmp_path / f"ext{bits}.wav"
        f.write_bytes(_wav_bytes(np.array([[lo, hi], [0, -1]]), 2, bits, 1))
        w, _ = pkg.wav_load(f)
        assert np.allclose(w[:, 0], [lo / div, hi / div], atol=1e-7) and w[0, 1] == 0
    # refused, not misread: 8-bit PCM, 64-bit float, 3 channels
    import pytest
    for smp, ch, bits, tag in ((np.zeros((4, 2)), 2, 8, 1), (np.zeros((4, 2)), 2, 64, 3), (np.zeros((4, 3)), 3, 16, 1)):
        f = tmp_path / "bad.wav"
        body = np.zeros(4 * ch * (bits // 8), np.uint8).tobytes()
        import struct
        fmt = struct.pack("<HHIIHH", tag, ch, 44100, 44100 * (bits // 8) * ch, (bits // 8) * ch, bits)
        f.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<I", 16) + fmt + b"data" +
                      struct.pack("<I", len(body)) + body)
        with pytest.raises(pkg.HostError):
            pkg.wav_load(f)


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
