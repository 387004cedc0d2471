"""Parity of the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

Tolerances (fp32 everywhere; BASELINE.json: "within a stated fp32 magnitude-spectrogram tolerance",
"SDR within +-0.05 dB"):
  STFT / magnitudes / network activations / target magnitudes : relative L2 <= 2e-5
  Wiener output spectrograms                                   : relative L2 <= 2e-4
  waveforms                                                    : max |diff| <= 1e-4 (the reference's
                                                                 own NEAR_TOLERANCE, test_dsp.cpp:7)
  SDR of HIP output measured against the oracle output         : >= 80 dB  (+-0.05 dB needs ~45 dB)
"""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import rel_l2, sdr_db

sys.path.insert(0, str(Path(__file__).parent))
import stagecheck  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"

TOL_STAGE, TOL_Y, TOL_WAVE, MIN_SDR = 2e-5, 2e-4, 1e-4, 80.0
# PARITY bounds above (TOL_WAVE = the reference's own 1e-4 of test_dsp.cpp).  REGRESSION bounds beside them (VERDICT round 2,
# weak #13): what is measured is ~1e-6 per stage (rel L2), ~3e-5 on the Wiener output (the EM step amplifies rounding
# noise ~100x) and <= 2e-6 on the waveforms, so a change that costs a decimal digit must not pass as "within parity".
REG_STAGE, REG_Y, REG_WAVE = 5e-6, 3e-5, 1e-5  # (REG_Y: round 5, 1e-4 -> 3e-5: measured <= 1.2e-5 since the filter is applied as v R (Cxx^-1 x))


def _check_report(rep):
    for seg, r in rep.items():
        for k, v in r.items():
            if k.startswith(("spec", "mix_mag", "x", "fc1", "lstm[", "fc2", "mask", "target_mag", "state")):
                assert v < TOL_STAGE, (seg, k, v)
                assert v < REG_STAGE, ("regression bound", seg, k, v)
            elif k.startswith("y["):
                assert v < TOL_Y, (seg, k, v)
                assert v < REG_Y, ("regression bound", seg, k, v)
            elif k.startswith("wave_maxabs"):
                assert v < TOL_WAVE, (seg, k, v)
                assert v < REG_WAVE, ("regression bound", seg, k, v)


@pytest.fixture(scope="module")
def small(pkg, po, model_small):
    path, om, targets = model_small
    N = 16 * 1024
    eng = pkg.Engine(targets, 128, N)
    yield eng, om, N
    eng.close()


def test_stage_parity_small_persistent_two_segments():
    rep = stagecheck.stage_report(128, 16 * 1024, segments=2, verbose=False)
    assert rep[0]["persistent"] is True
    _check_report(rep)


def test_stage_parity_small_stepwise():
    rep = stagecheck.stage_report(128, 16 * 1024, flags=0x10, segments=2, verbose=False)
    assert rep[0]["persistent"] is False
    _check_report(rep)


def test_stage_parity_umxl_hidden_64_frames():
    """hidden=1024 (UMX-L shapes: 256 LSTM workgroups, 64 weights per lane) on 65 frames."""
    rep = stagecheck.stage_report(1024, 64 * 1024, segments=2, verbose=False)
    assert rep[0]["persistent"] is True
    _check_report(rep)


@pytest.mark.parametrize("hidden", [256, 512])
def test_stage_parity_other_hidden_sizes(hidden):
    """hidden=512 is the reference's umxhq / umx models (model.cpp:109-114 reads the size from the file);
    256 exercises the remaining LSTM register layout (16 weights per lane)."""
    rep = stagecheck.stage_report(hidden, 32 * 1024, segments=2, verbose=False)
    assert rep[0]["persistent"] is True
    _check_report(rep)


def test_persistent_and_stepwise_lstm_are_bitwise_identical(pkg, small):
    eng, _, N = small
    wave = pkg.ggml.synth_audio(N, 21)
    eng.stream_reset()
    a = eng.infer_segment(wave)
    sa = eng.stream_get()
    assert eng.lstm_was_persistent()
    assert eng.lstm_mode() == 2  # census found every chain on one XCD
    for flag, mode in ((pkg.FLAG_LSTM_STEPWISE, 0), (pkg.FLAG_LSTM_FORCE_SAFE, 1)):
        eng.stream_reset()
        b = eng.infer_segment(wave, flag)
        sb = eng.stream_get()
        assert eng.lstm_mode() == mode
        assert (sa == sb).all()
        assert all((a[t] == b[t]).all() for t in range(4))


def test_umxl_persistent_modes_and_stepwise_are_bitwise_identical(pkg, tmp_path):
    """hidden=1024: the DPP-rotation kernel (intra-XCD and sc1 hand-off) and the per-step driver walk k
    in the same order, so all three give the same bits; so do the precise-activation variants."""
    H, N = 1024, 48 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=23), H, compress=False)
    eng = pkg.Engine.from_file(path, N)
    wave = pkg.ggml.synth_audio(N, 77)
    res = {}
    for name, flags in (("fast", 0), ("safe", pkg.FLAG_LSTM_FORCE_SAFE), ("step", pkg.FLAG_LSTM_STEPWISE)):
        for prec in (0, pkg.FLAG_PRECISE_ACT):
            eng.stream_reset()
            out = eng.infer_segment(wave, flags | prec)
            res[(name, prec)] = (out, eng.stream_get(), eng.lstm_mode())
    assert res[("fast", 0)][2] == 2 and res[("safe", 0)][2] == 1 and res[("step", 0)][2] == 0
    for prec in (0, pkg.FLAG_PRECISE_ACT):
        ref_out, ref_state, _ = res[("fast", prec)]
        for name in ("safe", "step"):
            out, state, _ = res[(name, prec)]
            assert (state == ref_state).all(), (name, prec)
            assert all((out[t] == ref_out[t]).all() for t in range(4)), (name, prec)
    # the two activation flavours differ only at the 1e-7 level
    a, b = res[("fast", 0)][0], res[("fast", pkg.FLAG_PRECISE_ACT)][0]
    assert 0 < max(np.abs(a[t] - b[t]).max() for t in range(4)) < 1e-5
    eng.close()


def test_pipelined_segments_equal_one_at_a_time(pkg, tmp_path):
    """Queuing segments back to back (device-pointer API, no sync in between) runs them through the two pipeline
    slots as an exact wavefront and must give the same bits as running them one at a time, including the carried
    LSTM state (F3)."""
    import torch
    torch.zeros(1).cuda()  # let torch initialise HIP before the engine's own streams exist
    H, N, NSEG = 1024, 40 * 1024, 6
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=29), H, compress=False)
    waves = [pkg.ggml.synth_audio(N, 200 + i) for i in range(NSEG)]
    eng = pkg.Engine.from_file(path, N)
    # (a) one at a time (host API syncs after every segment)
    eng.stream_reset()
    serial = [eng.infer_segment(w) for w in waves]
    s_state = eng.stream_get()
    # (b) back to back through the device-pointer API
    eng.stream_reset()
    ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
    outs = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in range(NSEG)]
    torch.cuda.synchronize()
    for i in range(NSEG):
        eng.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]])
    eng.sync()
    p_state = eng.stream_get()
    assert eng.lstm_mode() == 2
    for i in range(NSEG):
        for t in range(4):
            got = outs[i][t].cpu().numpy().reshape(N, 2).T
            assert (got == serial[i][t]).all(), (i, t)
    assert (s_state == p_state).all()
    eng.close()


def test_phased_segment_equals_whole_segment(pkg, small):
    """umx_hip_segment_begin / _lstm_layer x3 / _end (the cut points of the multi-GPU carry mode) run the
    same kernels as umx_hip_infer_segment: same bits, same carried state; per-layer get/set round-trips."""
    eng, om, N = small
    waves = [pkg.ggml.synth_audio(N, 300 + i) for i in range(3)]
    eng.stream_reset()
    whole = [eng.infer_segment(w) for w in waves]
    st_whole = eng.stream_get()
    eng.stream_reset()
    phased = []
    for w in waves:
        eng.segment_begin(w)
        for l in range(3):
            a = eng.stream_get_layer(l)
            eng.stream_set_layer(l, a)  # what another GPU would have sent
            eng.segment_lstm_layer(l)
        phased.append(eng.segment_end())
    st_phased = eng.stream_get()
    for i in range(3):
        for t in range(4):
            assert (whole[i][t] == phased[i][t]).all(), (i, t)
    assert (st_whole == st_phased).all()
    # the per-layer view is a slice of the whole state: [target][layer][dir][h|c][Hl]
    Hl = 64
    full = st_whole.reshape(4, 3, 2, 2, Hl)
    for l in range(3):
        assert (eng.stream_get_layer(l).reshape(4, 2, 2, Hl) == full[:, l]).all()
    # protocol errors are reported, not executed
    eng.segment_begin(waves[0])
    with pytest.raises(RuntimeError):
        eng.segment_lstm_layer(1)
    with pytest.raises(RuntimeError):
        eng.infer_segment(waves[0])
    for l in range(3):
        eng.segment_lstm_layer(l)
    eng.segment_end()
    eng.stream_reset()


def test_carry_mode_two_processes_one_gpu(pkg, model_small, tmp_path):
    """SURVEY 8e "carry" mode: a track's segments alternate between two ranks, each layer's (h, c) travels
    point to point; the result must be bit-identical to one engine running split_inference."""
    import socket
    import subprocess
    path, om, targets = model_small
    N, L, seed = 16 * 1024, int(16 * 1024 * 3.4), 77
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(root / "tests" / "carry_worker.py"), path, str(tmp_path), str(N), str(L), str(seed)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    carry = np.load(tmp_path / "carry.npy")
    eng = pkg.Engine(targets, 128, N)
    one = np.stack(pkg.split_inference(pkg.engine_backend(eng), pkg.ggml.synth_audio(L, seed), N))
    eng.close()
    assert carry.shape == one.shape
    assert (carry == one).all()


def test_quantised_resident_weights_are_bitwise_identical(pkg, model_small, tmp_path):
    """BASELINE config 5: u8/u16 matrices stay in HBM, a third of the weight memory.  With u8_dequant they are
    dequantised per element in the GEMM B-tile staging and the LSTM W_hh register load (q*scale+offset,
    model.cpp:610-616): same bits as dequantising at load time; persistent and per-step LSTM drivers, small and
    UMX-L-sized hidden.  The default treats u8 GEMM weights as exact bf16 integers and applies the affine map to
    the sum (three matrix-core products instead of six): equal to that within fp32 rounding, not bitwise."""
    path, om, targets = model_small
    N = 16 * 1024
    waves = [pkg.ggml.synth_audio(N, 400 + i) for i in range(2)]
    ref = pkg.Engine(targets, 128, N, quantised_resident=False)
    qr = pkg.Engine(targets, 128, N, u8_dequant=True)
    qx = pkg.Engine(targets, 128, N)  # the default
    assert qx.weight_bytes() * 1.8 < ref.weight_bytes()  # exact integer planes (1 or 2 per matrix) + u8 W_hh against 3 planes + fp32 W_hh
    for flags in (0, pkg.FLAG_LSTM_STEPWISE):
        ref.stream_reset()
        qr.stream_reset()
        qx.stream_reset()
        for w in waves:
            a, b, c = ref.infer_segment(w, flags), qr.infer_segment(w, flags), qx.infer_segment(w, flags)
            for t in range(4):
                assert (a[t] == b[t]).all(), (flags, t)
                assert 0 < np.abs(a[t] - c[t]).max() < 1e-5, (flags, t)  # both sit ~1e-6 from the oracle
        assert (ref.stream_get() == qr.stream_get()).all()
        assert rel_l2(qx.stream_get(), ref.stream_get()) < 2e-5
    qx.close()
    # fp32 views cannot stay quantised: the flag is then a no-op, not an error
    f32 = pkg.Engine(targets, 128, N, quantised=False)
    assert f32.weight_bytes() == ref.weight_bytes()
    for e in (ref, qr, f32):
        e.close()
    H, N = 1024, 24 * 1024
    p = str(tmp_path / "m.bin")
    pkg.ggml.write_model(p, pkg.ggml.synth_weights(H, seed=31), H, compress=False)
    ref, qr = pkg.Engine.from_file(p, N, quantised_resident=False), pkg.Engine.from_file(p, N, u8_dequant=True)
    assert 600e6 < ref.weight_bytes() < 640e6
    staged = pkg.Engine.from_file(p, N, gemm="bf16x3")  # round 1's kernels keep the file's u8 / u16 bytes in HBM
    assert 130e6 < staged.weight_bytes() < 150e6
    staged.close()
    with pytest.raises(pkg.UmxError):  # the fp32-MFMA flavour was removed in round 3: asking for it is an error, not a silent switch
        pkg.Engine.from_file(p, N, gemm="f32")
    w = pkg.ggml.synth_audio(N, 410)
    a, b = ref.infer_segment(w), qr.infer_segment(w)
    assert qr.lstm_mode() == 2
    for t in range(4):
        assert (a[t] == b[t]).all()
    ref.close()
    qr.close()


def test_mixed_weight_dtypes_across_targets(pkg, po, model_small):
    """A GEMM launch covers the four targets with one kernel instantiation, so a matrix may stay quantised only if
    EVERY target hands it over quantised: here target 1 brings fp32 views of fc1 / fc3 / one W_ih while the others
    bring the file's u8 / u16 -- the engine must expand those matrices for all targets (not read u8 bytes as bf16
    planes) and still sit on the oracle."""
    path, om, targets = model_small
    N = 16 * 1024
    mixed = [dict(t) for t in targets]
    for name in ("fc1.weight", "fc3.weight", "lstm.weight_ih_l1_reverse", "lstm.weight_hh_l2"):
        mixed[1][name] = np.ascontiguousarray(mixed[1][name]["f32"])
    eng = pkg.Engine(mixed, 128, N)
    full = pkg.Engine(targets, 128, N)
    assert eng.weight_bytes() > full.weight_bytes()
    wave = pkg.ggml.synth_audio(N, 430)
    got = eng.infer_segment(wave)
    ref, _ = po.umx_inference(om, wave)
    for t in range(4):
        assert np.abs(got[t] - ref[t]).max() < TOL_WAVE
    eng.close()
    full.close()


def test_device_resident_track_equals_host_split_and_shift(pkg, small):
    """umx_hip_split_inference / umx_hip_shift_inference (track in HBM, pipelined segments, overlap-add on
    the device) against the C++ host drivers of umx_host.h calling umx_hip_infer_segment per segment:
    the same bits, for a multi-segment track with a ragged tail, a sub-segment track and a 1-sample track."""
    eng, om, N = small
    be = pkg.engine_backend(eng)
    for L, seed in ((int(N * 3.4), 500), (N // 3, 501), (N, 502), (1, 503)):
        wave = pkg.ggml.synth_audio(max(L, 16), seed)[:, :L]
        host = pkg.split_inference(be, wave, N)
        dev = eng.separate(wave)
        for t in range(4):
            assert (host[t] == dev[t]).all(), ("split", L, t)
    wave = pkg.ggml.synth_audio(int(N * 2.2), 504)
    host = pkg.shift_inference(be, wave, N, offset=4033)
    dev = eng.separate(wave, shift_offset=4033)
    for t in range(4):
        assert (host[t] == dev[t]).all(), ("shift", t)
    with pytest.raises(RuntimeError):
        eng.separate(wave, shift_offset=22050)
    # the per-segment API still works afterwards and starts from the state the track left
    eng.stream_reset()


def test_fused_wiener_istft_equals_the_unfused_kernels_bitwise(pkg, model_small, monkeypatch):
    """csrc/wiener_istft.h (gains + filter + inverse STFT frame in one kernel: the track-batched default) must give the
    bits of the separate filter kernel (wiener_apply_kernel: the same per-bin functions, one bin per thread) followed by the separate
    inverse STFT (UMX_WIENER=stats4: the single-track default), with and without the EM step (BASELINE config 2),
    including the y tap."""
    import torch
    torch.zeros(1).cuda()
    path, om, targets = model_small
    # 41 frames: five runs of nine -- frames kept at a run's start, the next frame's inputs requested under the overlap-add, interior frames
    # (window sum-square from LDS) and the three edge frames at either end, a last STFT run of one frame; 6 frames: one run, no interior frame
    for N in (40 * 1024, 5 * 1024):
        wave = pkg.ggml.synth_audio(N, 77)
        for flags in (pkg.FLAG_DEBUG_TAPS, pkg.FLAG_DEBUG_TAPS | pkg.FLAG_NO_WIENER):
            res = {}
            for mode in ("stats4", "fused"):
                monkeypatch.setenv("UMX_WIENER", mode)
                eng = pkg.Engine(targets, 128, N)
                res[mode] = (eng.infer_segment(wave, flags), [eng.tap("y", t) for t in range(4)])
                eng.close()
            for t in range(4):
                assert np.array_equal(res["fused"][0][t], res["stats4"][0][t]), (N, flags, t)
                assert np.array_equal(res["fused"][1][t], res["stats4"][1][t]), (N, flags, t)


def test_wiener_bin_arithmetic_against_a_reference_order_float64_restatement(pkg):
    """both filter kernels call wiener_bin_setup / wiener_bin_apply (csrc/wiener_kernels.h), so the fused-vs-unfused
    test cannot see an error in that shared per-bin math, and since round 4 it forms y_j = v_j R_j (Cxx^-1 x) instead of the
    reference's (v_j R_j Cxx^-1) x.  Here the device functions run on given bins (umx_hip_debug_wiener_bins) against numpy
    float64 in the REFERENCE's operation order (wiener.cpp:187-202 PSD with F5, :301-325 Cxx with the 4x regularisation F6,
    :54-84 inverse, :339-376 gain, :381-400 apply, :408-422 rescale): well-conditioned bins to 1e-5 of the bin's largest output, all-zero
    masks (Cxx = 4 sqrt(eps) I, v = 0: exact zeros), a silent mixture bin (arg 0 = 0), a near-silent one (1e-20: below the range in which
    sqrt(re^2 + im^2) is accurate, treated as silent), and NaN in -> NaN out without touching its neighbours."""
    import ctypes as C
    rng = np.random.default_rng(5)
    n = 4096
    X = (rng.standard_normal((n, 2, 2)) * rng.uniform(0.01, 30.0, (n, 1, 1))).astype(np.float32)
    masks = rng.uniform(0.0, 1.5, (n, 4, 2)).astype(np.float32)
    A = rng.standard_normal((n, 4, 2, 2)) + 1j * rng.standard_normal((n, 4, 2, 2))
    Rm = A @ np.conj(np.swapaxes(A, -1, -2)) / 2 + 0.05 * np.eye(2)  # Hermitian positive definite, like sum y y^H / weight
    R = np.stack([Rm[..., 0, 0].real, Rm[..., 0, 1].real, Rm[..., 0, 1].imag, Rm[..., 1, 1].real], axis=-1).astype(np.float32)
    masks[0] = 0.0          # nothing of any source in this bin
    X[1] = 0.0              # a silent mixture bin
    X[2, 0, 0] = np.nan     # a poisoned bin
    X[3] = np.float32(1e-20) * np.array([[0.6, -0.8], [-0.28, 0.96]], np.float32)  # a near-silent bin: re^2 + im^2 underflows
    max_abs = np.float32(max(1.0, 30.0 / 10.0))
    y = np.zeros((n, 4, 2, 2), np.float32)
    fp = C.POINTER(C.c_float)
    import torch
    torch.zeros(1).cuda()  # a current device
    rc = pkg.hip_lib().umx_hip_debug_wiener_bins(n, X.ctypes.data_as(fp), masks.ctypes.data_as(fp), R.ctypes.data_as(fp), C.c_float(float(max_abs)), y.ctypes.data_as(fp))
    assert rc == 0
    got = y[..., 0] + 1j * y[..., 1]  # (n, 4, 2)
    # ---- float64, the reference's order, from the float32 inputs
    Xc = X[..., 0].astype(np.float64) + 1j * X[..., 1].astype(np.float64)  # (n, 2)
    mag = np.abs(Xc)
    ph = np.where(mag > 0, Xc / np.where(mag > 0, mag, 1.0), 1.0)
    Rd = R.astype(np.float64)
    Rfull = np.empty((n, 4, 2, 2), np.complex128)
    Rfull[..., 0, 0], Rfull[..., 1, 1] = Rd[..., 0], Rd[..., 3]
    Rfull[..., 0, 1] = Rd[..., 1] + 1j * Rd[..., 2]
    Rfull[..., 1, 0] = Rd[..., 1] - 1j * Rd[..., 2]
    M = float(max_abs)
    x = Xc / M
    ys = (masks.astype(np.float64) * mag[:, None, :]) * ph[:, None, :] / M  # (n, 4, 2): polar(mag_j, arg X) / max_abs
    v = 0.5 * ((ys.real + ys.imag) ** 2).sum(axis=-1)  # F5
    Cxx = (np.sqrt(1e-10) * np.eye(2)[None, None] + v[..., None, None] * Rfull).sum(axis=1)  # F6: once per source
    inv = np.linalg.inv(np.where(np.isfinite(Cxx), Cxx, np.eye(2)))
    G = v[..., None, None] * (Rfull @ inv[:, None])  # wiener.cpp:339-376
    ref = np.einsum("nsab,nb->nsa", G, x) * M
    ok = np.isfinite(Xc).all(axis=-1)
    scale = np.abs(ref[ok]).max(axis=(1, 2), keepdims=True) + 1e-30
    err = np.abs(got[ok] - ref[ok]) / scale
    worst = float(err[3:].max())
    assert worst < 1e-5, worst  # measured ~2e-6: fp32 evaluation of a 2 x 2 solve with condition numbers up to ~1e2
    assert (got[0] == 0).all()                       # v = 0: exact zeros
    assert (got[1] == 0).all() or float(np.abs(got[1]).max()) == 0.0  # x = 0
    assert not np.isfinite(got[2]).all()             # NaN in -> NaN out
    assert np.isfinite(got[3:]).all()
    # |X| below 1e-18 (common.h: the squares leave the normal range) counts as a silent bin of phase 0: the absolute difference to
    # the reference's polar(mag, arg X) is of the size of the bin itself
    assert float(np.abs(got[3] - ref[3]).max()) < 1e-16, np.abs(got[3] - ref[3]).max()


def test_gemm_flavours_agree(pkg, po, model_small, tmp_path):
    """The dense stack runs on the 16-bit matrix cores with split operands and fp32 accumulation: gemm="bf16x3" splits into
    three bf16 terms while it stages every tile (csrc/gemm_bf16x3.h, the single-track default), gemm="planes" consumes
    operands pre-split into fp16 planes by LDS-DMA (csrc/gemm_planes.h, the track-batched default).  Both must sit within
    the same distance of the oracle, agree with each other to fp32 rounding, and -- for either flavour -- queuing segments back to back must give the bits of one segment at a time (co-residency of bf16
    MFMA waves with other kernels' waves is what this guards)."""
    import torch
    torch.zeros(1).cuda()
    path, om, targets = model_small
    N = 16 * 1024
    waves = [pkg.ggml.synth_audio(N, 600 + i) for i in range(2)]
    state = po.stream_state(128)
    ref = [po.umx_inference(om, w, n_buf=N, state=state)[0] for w in waves]
    outs = {}
    for gemm in ("bf16x3", "planes"):
        eng = pkg.Engine(targets, 128, N, gemm=gemm)
        outs[gemm] = [eng.infer_segment(w) for w in waves]
        eng.close()
        for i in range(2):
            for t in range(4):
                assert np.abs(outs[gemm][i][t] - ref[i][t]).max() < TOL_WAVE
    for i in range(2):
        for t in range(4):
            assert np.abs(outs["bf16x3"][i][t] - outs["planes"][i][t]).max() < 1e-5
    # UMX-L width, pipelined == serial, both flavours
    H, N, NSEG = 1024, 40 * 1024, 6
    p = str(tmp_path / "m.bin")
    pkg.ggml.write_model(p, pkg.ggml.synth_weights(H, seed=33), H, compress=False)
    waves = [pkg.ggml.synth_audio(N, 610 + i) for i in range(NSEG)]
    ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
    for gemm in ("bf16x3", "planes"):
        eng = pkg.Engine.from_file(p, N, gemm=gemm)
        eng.stream_reset()
        serial = [eng.infer_segment(w) for w in waves]
        eng.stream_reset()
        o = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in range(NSEG)]
        torch.cuda.synchronize()
        for i in range(NSEG):
            eng.infer_segment_device(ins[i].data_ptr(), N, [x.data_ptr() for x in o[i]])
        eng.sync()
        for i in range(NSEG):
            for t in range(4):
                assert (o[i][t].cpu().numpy().reshape(N, 2).T == serial[i][t]).all(), (gemm, i, t)
        eng.close()


def test_full_size_pipelined_equals_serial(pkg, tmp_path):
    """UMX-L at the real segment length (T = 2584): four segments queued back to back -- two LSTM grids, GEMM blocks
    (bf16 MFMA), FFT and Wiener workgroups all sharing CUs for milliseconds -- against the same four one at a time.
    Bitwise.  This is the guard for cross-kernel interference."""
    import torch
    torch.zeros(1).cuda()
    H, N, NSEG = 1024, pkg.SEGMENT_SAMPLES, 4
    p = str(tmp_path / "m.bin")
    pkg.ggml.write_model(p, pkg.ggml.synth_weights(H, seed=37), H, compress=False)
    eng = pkg.Engine.from_file(p, N)
    wave = pkg.ggml.synth_audio(N + 3 * 4096, 700)
    chunks = [np.ascontiguousarray(wave[:, i * 4096:i * 4096 + N]) for i in range(NSEG)]
    eng.stream_reset()
    serial = [eng.infer_segment(c) for c in chunks]
    ins = [torch.from_numpy(np.ascontiguousarray(c.T).ravel()).cuda() for c in chunks]
    o = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in range(NSEG)]
    eng.stream_reset()
    torch.cuda.synchronize()
    for i in range(NSEG):
        eng.infer_segment_device(ins[i].data_ptr(), N, [x.data_ptr() for x in o[i]])
    eng.sync()
    assert eng.lstm_mode() == 2
    for i in range(NSEG):
        for t in range(4):
            assert (o[i][t].cpu().numpy().reshape(N, 2).T == serial[i][t]).all(), (i, t)
    eng.close()


def test_error_behaviour_of_the_c_abi(pkg, model_small, small):
    """Errors come back as status codes + umx_hip_last_error, never as exit() (the reference prints and exits,
    model.cpp:59-64, dsp.cpp:27-44); a failed call leaves the context usable."""
    import ctypes as C
    eng, om, N = small
    lib = eng.lib
    fp = C.POINTER(C.c_float)
    outs = [np.empty(2 * N, np.float32) for _ in range(4)]
    arr = (fp * 4)(*[o.ctypes.data_as(fp) for o in outs])
    a = np.zeros(2 * (N + 1), np.float32)
    for n in (0, -3, N + 1):  # chunk longer than the stft buffer / empty
        assert lib.umx_hip_infer_segment(eng.h, a.ctypes.data_as(fp), n, arr, 0) == pkg.ERR_ARG
        assert b"segment" in lib.umx_hip_last_error(eng.h).lower()
    assert lib.umx_hip_infer_segment(eng.h, None, 16, arr, 0) == pkg.ERR_ARG
    assert lib.umx_hip_stream_set(eng.h, None) == pkg.ERR_ARG
    assert lib.umx_hip_stream_get_layer(eng.h, 3, a.ctypes.data_as(fp)) == pkg.ERR_ARG
    assert lib.umx_hip_segment_lstm_layer(eng.h, 0) == pkg.ERR_ARG  # no phased segment open
    assert lib.umx_hip_split_inference(eng.h, a.ctypes.data_as(fp), 0, arr, 0, None, None) == pkg.ERR_ARG
    # creation: wrong hidden size, missing tensor, bad device
    path, _, targets = model_small
    with pytest.raises(pkg.UmxError) as e:
        pkg.Engine(targets, 100, N)
    assert e.value.code == pkg.ERR_ARG
    with pytest.raises(pkg.UmxError) as e:
        pkg.Engine(targets, 128, N, device=99)
    assert e.value.code == pkg.ERR_ARG
    broken = [dict(t) for t in targets]
    del broken[2]["fc2.weight"]
    with pytest.raises((pkg.UmxError, KeyError)):
        pkg.Engine(broken, 128, N)
    # track lanes x segment length beyond the 32-bit addressing of the plane GEMMs' operands: refused at create, before any large
    # allocation (48 lanes x 3,900 frames: the second fp16 plane of fc1's operand would start past 2^31 bytes)
    with pytest.raises(pkg.UmxError) as e:
        pkg.Engine(targets, 128, 3900 * 1024, tracks=48)
    assert e.value.code == pkg.ERR_ARG and "32-bit" in str(e.value)
    # still alive
    w = pkg.ggml.synth_audio(N, 800)
    eng.stream_reset()
    assert all(np.isfinite(s).all() for s in eng.infer_segment(w))


def test_short_chunk_ragged_last_segment(pkg, po, small):
    """n < segment_samples: T stays n_buf/1024+1, the tail is zeros, outputs are (2,n) (a3, a11)."""
    eng, om, N = small
    for n in (1, 777, N - 1):
        wave = pkg.ggml.synth_audio(max(n, 2), 5)[:, :n]
        eng.stream_reset()
        got = eng.infer_segment(wave)
        ref, _ = po.umx_inference(om, wave, n_buf=N)
        for t in range(4):
            assert got[t].shape == (2, n)
            assert np.abs(got[t] - ref[t]).max() < TOL_WAVE


def test_silence_and_full_scale_noise(pkg, po, small):
    """Edge inputs of SURVEY 8d: all-zero audio (0/0 guards: arg(0)=0, eps in the Wiener weight) and
    full-scale white noise."""
    eng, om, N = small
    eng.stream_reset()
    z = np.zeros((2, N), np.float32)
    got = eng.infer_segment(z)
    ref, _ = po.umx_inference(om, z)
    for t in range(4):
        assert np.isfinite(got[t]).all() and np.abs(got[t] - ref[t]).max() < TOL_WAVE
    noise = np.random.default_rng(0).uniform(-1, 1, (2, N)).astype(np.float32)
    eng.stream_reset()
    got = eng.infer_segment(noise)
    ref, _ = po.umx_inference(om, noise)
    for t in range(4):
        assert sdr_db(ref[t], got[t]) > MIN_SDR


def test_stream_state_get_set_and_reset(pkg, po, small):
    """The streaming LSTM state (F3): segment 2 depends on segment 1; get/set reproduces it exactly
    (this is what a checkpoint or a multi-GPU hand-off uses); reset returns to the first-segment result."""
    eng, om, N = small
    w1, w2 = pkg.ggml.synth_audio(N, 31), pkg.ggml.synth_audio(N, 32)
    eng.stream_reset()
    eng.infer_segment(w1)
    st = eng.stream_get()
    b = eng.infer_segment(w2)
    eng.stream_reset()
    c = eng.infer_segment(w2)  # no carry: must differ
    assert np.abs(b[3] - c[3]).max() > 1e-7
    eng.stream_set(st)
    d = eng.infer_segment(w2)  # restored carry: bitwise the same
    assert all((b[t] == d[t]).all() for t in range(4))
    ost = po.stream_state(128)
    po.umx_inference(om, w1, state=ost)
    assert rel_l2(st, ost) < TOL_STAGE


def test_flags_no_wiener_and_skip_targets(pkg, po, small):
    eng, om, N = small
    wave = pkg.ggml.synth_audio(N, 41)
    eng.stream_reset()
    got = eng.infer_segment(wave, pkg.FLAG_NO_WIENER)  # BASELINE config 2
    ref, _ = po.umx_inference(om, wave, flags=1)
    for t in range(4):
        assert np.abs(got[t] - ref[t]).max() < TOL_WAVE
    skip = pkg.FLAG_SKIP_TARGET(0) | pkg.FLAG_SKIP_TARGET(1) | pkg.FLAG_SKIP_TARGET(2)  # config 1: vocals only
    eng.stream_reset()
    got = eng.infer_segment(wave, skip)
    ref, _ = po.umx_inference(om, wave, flags=skip)
    for t in range(4):
        assert np.abs(got[t] - ref[t]).max() < TOL_WAVE
    assert np.abs(got[0]).max() < 1e-6 and np.abs(got[3]).max() > 1e-3


def test_stft_istft_roundtrip_on_device_with_an_identity_mask_model(pkg):
    """The reference's round-trip property (test_dsp.cpp:41-114: STFT -> iSTFT returns the input within 1e-4) through
    the whole engine: a model whose mask is exactly 1 (fc3 = 0, bn3 mean = bias = 0, output_mean = 1, so
    relu(0 * scale + 1) = 1, inference.cpp:143-166) with the Wiener step off (y = mask |X| e^{i arg X} = X,
    wiener.cpp:96-109) makes every stem the device's iSTFT(STFT(input)).  Size-independent: also at the production
    segment length."""
    H = 128
    W = pkg.ggml.synth_weights(H, seed=61)
    for t in W:
        t["fc3.weight"] = np.zeros_like(t["fc3.weight"])
        t["bn3.running_mean"] = np.zeros_like(t["bn3.running_mean"])
        t["bn3.bias"] = np.zeros_like(t["bn3.bias"])
        t["output_mean"] = np.ones_like(t["output_mean"])
    for N, n in ((16 * 1024, 16 * 1024), (16 * 1024, 9000), (pkg.SEGMENT_SAMPLES, pkg.SEGMENT_SAMPLES)):
        eng = pkg.Engine(W, H, N, quantised=False)
        wave = pkg.ggml.synth_audio(n, 51)
        got = eng.infer_segment(wave, pkg.FLAG_NO_WIENER | pkg.FLAG_DEBUG_TAPS)
        assert np.abs(eng.tap("mask", 2) - 1.0).max() == 0.0
        for t in range(4):
            assert np.abs(got[t] - wave).max() < 1e-4, (N, n, t)  # test_dsp.cpp:7 NEAR_TOLERANCE
        # taps: |spec| is the mix magnitude, x is its 1487-bin crop stacked L|R (inference.cpp:29,52-68)
        spec, mag, x = eng.tap("spec"), eng.tap("mix_mag"), eng.tap("x")
        assert np.abs(np.abs(spec) - mag).max() < 1e-6 * max(mag.max(), 1.0)
        assert (x[:, :1487] == mag[0, :, :1487]).all() and (x[:, 1487:2974] == mag[1, :, :1487]).all()
        assert (x[:, 2974:] == 0).all()
        eng.close()


def test_wiener_stems_sum_back_to_the_mix(pkg, small):
    """With the Wiener step on, the four estimates sum back to the mixture -- approximately: the gains G_j =
    v_j R_j Cxx^-1 sum to I - 4 sqrt(eps) Cxx^-1 (F6), and the (Re+Im)^2 PSD of F5 applies to the update, not to this
    identity.  SURVEY 8c item 2 measured 0.2 % on random data; with the random (untrained) masks here 2 % is asserted."""
    eng, _, N = small
    wave = pkg.ggml.synth_audio(N, 51)
    eng.stream_reset()
    got = eng.infer_segment(wave)
    assert np.abs(sum(got) - wave).max() / np.abs(wave).max() < 2e-2


def test_full_size_segment_vs_oracle():
    """BASELINE config 3 at full size: hidden 1024, one 60 s segment (T = 2584), 4 stems + Wiener,
    two consecutive segments (streaming carry).  The oracle needs a many-core host (it takes seconds
    on the GPU box's 256 cores); every stage is compared, then SDR of HIP vs oracle."""
    rep = stagecheck.stage_report(1024, 2_646_000, segments=2, verbose=False)
    assert rep[0]["persistent"] is True
    _check_report(rep)


def test_full_size_config2_no_wiener_vs_oracle():
    """BASELINE config 2 at full size (hidden 1024, T = 2584): 4 stems, mixture phase instead of the Wiener EM."""
    rep = stagecheck.stage_report(1024, 2_646_000, flags=0x1, segments=1, verbose=False)
    assert rep[0]["persistent"] is True
    _check_report(rep)


def test_full_size_config1_vocals_only_vs_oracle():
    """BASELINE config 1 at full size: the vocals model only (targets 0-2 skipped: their magnitudes are zero), one
    60 s segment, against the oracle run the same way."""
    rep = stagecheck.stage_report(1024, 2_646_000, flags=0x700, segments=1, verbose=False)
    assert rep[0]["persistent"] is True
    assert "lstm[3]" in rep[0] and "lstm[0]" not in rep[0]
    _check_report(rep)


def test_shipped_track_end_to_end_sdr(pkg, po, tmp_path):
    """The reference's shipped test track (test/data/gspi_stereo.wav, 5.94 s) through the whole
    product path -- C++ wav reader, C++ ggml loader, C++ shift/split drivers, HIP engine with the
    production 60 s segment size -- against the oracle's shift_inference (F4 fixed on both sides).
    BASELINE: SDR within +-0.05 dB; here the HIP output is scored against the oracle output."""
    H = 128
    path = str(tmp_path / "m.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=17), H)
    wave, ch = pkg.wav_load(GOLD / "gspi_stereo.wav")
    assert ch == 2 and wave.shape == (2, 262144)
    hm = pkg.HostModel(path)
    eng = pkg.engine_from_host_model(hm, pkg.SEGMENT_SAMPLES)
    got = pkg.shift_inference(pkg.engine_backend(eng), wave, pkg.SEGMENT_SAMPLES, offset=4033)
    om = po.Model.load(path)
    ref = po.shift_inference(om, wave, pkg.SEGMENT_SAMPLES, 4033)
    for t in range(4):
        assert sdr_db(ref[t], got[t]) > MIN_SDR
        assert np.abs(got[t] - ref[t]).max() < TOL_WAVE
    eng.close()


def test_cli_end_to_end(pkg, po, tmp_path):
    """umx-cli <model file> <wav file> <out dir> (umx.cpp:26-97): writes target_{0..3}.wav."""
    import subprocess
    H = 128
    path = str(tmp_path / "m.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=17), H)
    out = tmp_path / "out"
    cli = Path(pkg.HERE) / "umx-cli"
    r = subprocess.run([str(cli), path, str(GOLD / "gspi_stereo.wav"), str(out)], capture_output=True, text=True,
                       env={**__import__("os").environ, "UMX_SHIFT_OFFSET": "4033"}, timeout=600)
    assert r.returncode == 0, r.stderr
    om = po.Model.load(path)
    wave, _ = pkg.wav_load(GOLD / "gspi_stereo.wav")
    ref = po.shift_inference(om, wave, pkg.SEGMENT_SAMPLES, 4033)
    for t in range(4):
        got, ch = pkg.wav_load(out / f"target_{t}.wav")
        assert ch == 2 and got.shape == wave.shape
        assert sdr_db(ref[t], got) > MIN_SDR
    bad = subprocess.run([str(cli), path], capture_output=True, text=True)
    assert bad.returncode == 1 and "Usage" in bad.stderr  # umx.cpp:28-33
