"""Track batching: B independent tracks through one context, their LSTM recurrence in ONE launch per
layer on the matrix cores (csrc/lstm_batch.h).  The reference is one track per process (umx.cpp:26-97); what must
hold is that every track of a batch gets what `umx_inference` (inference.cpp:12-207) would have given it alone:
  * against the oracle, per track, with the tolerances of test_gpu_parity.py;
  * bitwise independence from the batch: same bits whatever the lane, the companions and the batch size
    (B = 1, 4, 16 -- the B = 1 engine is created with the batched kernel), and for the per-step driver;
  * a lane that sits a call out keeps its streaming state.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL_STAGE, TOL_WAVE = 2e-5, 1e-4  # parity bounds (test_gpu_parity.py)
REG_STAGE, REG_Y, REG_WAVE = 5e-6, 3e-5, 1e-5  # regression bounds beside them: measured ~1e-6 / <= 1.2e-5 / <= 2e-6


def _oracle_track(po, om, hidden, waves, n_buf):
    """One track = consecutive segments with carried state through the oracle; -> ([outs per segment], state)."""
    st = po.stream_state(hidden)
    outs = []
    for w in waves:
        ref, _ = po.umx_inference(om, w, n_buf=n_buf, state=st)
        outs.append(ref)
    return outs, np.array(st, copy=True)


@pytest.mark.parametrize("hidden,frames,quantised", [(128, 16, True), (128, 16, False), (512, 24, True), (1024, 40, True)])
def test_batch_of_tracks_matches_oracle_per_track(pkg, po, tmp_path, hidden, frames, quantised):
    B, N, NSEG = 4, frames * 1024, 2
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(hidden, seed=41), hidden, compress=False)
    om = po.Model.load(path)
    eng = pkg.Engine.from_file(path, N, tracks=B, quantised=quantised)
    assert eng.lstm_is_batched()
    # different audio, and ragged lengths, per lane
    lens = [N, N - 777, N, 5000]
    waves = [[pkg.ggml.synth_audio(lens[b], 500 + 10 * b + s) for s in range(NSEG)] for b in range(B)]
    got = [eng.infer_batch([waves[b][s] for b in range(B)]) for s in range(NSEG)]
    assert eng.lstm_was_persistent()
    for b in range(B):
        ref, ref_state = _oracle_track(po, om, hidden, waves[b], N)
        for s in range(NSEG):
            for t in range(4):
                err = float(np.abs(got[s][b][t] - ref[s][t]).max())
                assert err < TOL_WAVE, (hidden, b, s, t, err)
        assert rel_l2(eng.track_stream_get(b), ref_state) < TOL_STAGE, (hidden, b)
    eng.close()


def test_track_bits_do_not_depend_on_lane_companions_or_batch_size(pkg, tmp_path):
    """UMX-L width.  The same track run (a) alone on a 1-track context with the batched kernel, (b) as lane 2 of a
    4-track batch, (c) as lane 13 of a 16-track batch among other audio, (d) with the per-step driver: identical
    stems and identical carried state, bit for bit."""
    H, N, NSEG = 1024, 24 * 1024, 2
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=43), H, compress=False)
    track = [pkg.ggml.synth_audio(N, 900 + s) for s in range(NSEG)]
    other = [pkg.ggml.synth_audio(N, 950 + i) for i in range(64)]
    results = []
    for B, lane, flags in ((1, 0, 0), (4, 2, 0), (16, 13, 0), (4, 1, pkg.FLAG_LSTM_STEPWISE), (4, 3, pkg.FLAG_LSTM_FORCE_SAFE),
                           (32, 5, 0), (32, 29, 0), (20, 17, pkg.FLAG_LSTM_STEPWISE), (24, 21, pkg.FLAG_LSTM_FORCE_SAFE),  # up to 32 lanes: one octet per workgroup
                           (48, 44, 0), (40, 35, pkg.FLAG_LSTM_STEPWISE),  # 33 .. 64: two octets per workgroup in turn
                           (64, 50, 0), (64, 3, 0), (56, 33, pkg.FLAG_LSTM_STEPWISE), (40, 19, pkg.FLAG_LSTM_FORCE_SAFE)):
        eng = pkg.Engine.from_file(path, N, tracks=B, lstm_batched=True)
        outs = []
        for s in range(NSEG):
            batch = [other[(i + s) % 64] for i in range(B)]
            batch[lane] = track[s]
            outs.append(eng.infer_batch(batch, flags)[lane])
        results.append((outs, eng.track_stream_get(lane), B, lane, flags))
        eng.close()
    ref_outs, ref_state = results[0][0], results[0][1]
    for outs, state, B, lane, flags in results[1:]:
        assert (state == ref_state).all(), (B, lane, flags)
        for s in range(NSEG):
            for t in range(4):
                assert (outs[s][t] == ref_outs[s][t]).all(), (B, lane, flags, s, t)


@pytest.mark.parametrize("flags", [0, 0x1, 0x700], ids=["config3", "config2_no_wiener", "config1_vocals_only"])
def test_gemm_tiles_that_straddle_track_lanes(pkg, po, tmp_path, flags):
    """Lanes follow each other every T rows in the lane-contiguous activation buffers, so the 128- / 256-row tiles of the
    plane GEMMs hold rows of two lanes whenever T is not a multiple of the tile (the production T = 2584 is not): 301
    frames, three lanes of different audio, one of them absent in the second call.  Per lane: the oracle's answer, the
    mask tap included (the fc3 epilogue maps rows to (lane, frame) itself), and the bits of the same track alone."""
    hidden, N, B, NSEG = 128, 300 * 1024, 3, 2
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(hidden, seed=53), hidden, compress=False)
    om = po.Model.load(path)
    waves = [[pkg.ggml.synth_audio(N - 1000 * b, 800 + 10 * b + s) for s in range(NSEG)] for b in range(B)]
    eng = pkg.Engine.from_file(path, N, tracks=B)
    assert eng.T == 301
    got, masks = [], []
    for s in range(NSEG):
        batch = [waves[b][s] for b in range(B)]
        if s == 1:
            batch[1] = None  # lane 1 sits the second call out
        got.append(eng.infer_batch(batch, pkg.FLAG_DEBUG_TAPS | flags))
        masks.append([None if batch[b] is None else eng.tap(f"mask#{b}", 3) for b in range(B)])
    eng.close()
    for b in range(B):
        st = po.stream_state(hidden)
        alone = pkg.Engine.from_file(path, N, tracks=1, lstm_batched=True, gemm="planes")
        for s in range(NSEG):
            if s == 1 and b == 1:
                continue
            ref, taps = po.umx_inference(om, waves[b][s], n_buf=N, state=st, flags=flags, want_taps=True)
            one = alone.infer_batch([waves[b][s]], flags)[0]
            for t in range(4):
                assert float(np.abs(got[s][b][t] - ref[t]).max()) < TOL_WAVE, (b, s, t)
                assert (got[s][b][t] == one[t]).all(), (b, s, t)
            assert rel_l2(masks[s][b], taps["mask"][3]) < TOL_STAGE, (b, s)
        alone.close()


def test_ping_pong_gemm_gives_the_bits_of_the_lock_step_kernel(pkg, tmp_path, monkeypatch):
    """csrc/gemm_planes_pp.h (eight waves of 128 x 64 in two groups half a trip apart, the default for one-plane / u8
    weights) against csrc/gemm_planes.h (sixteen waves of 64 x 64 in lock step): every accumulator sees the same
    sequence of matrix instructions, so the stems, the carried state and every tap must agree bit for bit -- with the
    ping-pong kernel forced on all four GEMMs (two-plane / u16 weights included: its half-tile phases), on none, and
    as shipped (= all four since the end of round 3).  Launches large enough for the 256 x 256 tiles in every GEMM (8 lanes x 900 frames, hidden 512)."""
    H, N, B = 512, 900 * 1024, 8
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=47), H, compress=False)
    waves = [[pkg.ggml.synth_audio(N - 333 * b, 1300 + 10 * b + s) for b in range(B)] for s in range(2)]
    res = {}
    for pp in ("15", "0", None):
        if pp is None:
            monkeypatch.delenv("UMX_GEMM_PP", raising=False)
        else:
            monkeypatch.setenv("UMX_GEMM_PP", pp)
        eng = pkg.Engine.from_file(path, N, tracks=B, quantised=True)
        outs = [eng.infer_batch(w) for w in waves]
        res[pp] = (outs, [eng.track_stream_get(b) for b in range(B)])
        eng.close()
    for pp in ("0", None):
        for s in range(2):
            for b in range(B):
                for t in range(4):
                    assert (res[pp][0][s][b][t] == res["15"][0][s][b][t]).all(), (pp, s, b, t)
        for b in range(B):
            assert (res[pp][1][b] == res["15"][1][b]).all(), (pp, b)


def test_persistent_gemm_gives_the_bits_of_the_ping_pong_kernel(pkg, tmp_path, monkeypatch):
    """csrc/gemm_planes_ps.h (round 6: one workgroup per CU walks the tiles of all four targets, a tile's epilogue inside the first
    trip of the next tile, tables of row / column vectors in LDS) against csrc/gemm_planes_pp.h (one tile per workgroup): every
    accumulator sees the same matrix instructions in the same order and every element the same scalar epilogue, so stems, carried
    state and the mask tap agree bit for bit -- with the persistent kernel on all four GEMMs, on none, and as shipped.  16 lanes x 900
    frames, hidden 512: more 256 x 256 tiles than CUs in every GEMM, tiles that straddle two lanes and the M padding in fc3."""
    H, N, B = 512, 900 * 1024, 16
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=53), H, compress=False)
    waves = [[pkg.ggml.synth_audio(N - 333 * b, 1500 + 10 * b + s) for b in range(B)] for s in range(2)]
    res = {}
    for ps in ("15", "0", None):
        if ps is None:
            monkeypatch.delenv("UMX_GEMM_PS", raising=False)
        else:
            monkeypatch.setenv("UMX_GEMM_PS", ps)
        eng = pkg.Engine.from_file(path, N, tracks=B, quantised=True)
        outs = [eng.infer_batch(w, pkg.FLAG_DEBUG_TAPS) for w in waves]
        taps = {f"{name}#{b}": eng.tap(f"{name}#{b}", 3) for b in (0, B - 1) for name in ("fc1", "fc2", "mask")}
        res[ps] = (outs, [eng.track_stream_get(b) for b in range(B)], taps)
        eng.close()
    for ps in ("0", None):
        for k, v in res["15"][2].items():
            assert (res[ps][2][k] == v).all(), (ps, k)
        for s in range(2):
            for b in range(B):
                for t in range(4):
                    assert (res[ps][0][s][b][t] == res["15"][0][s][b][t]).all(), (ps, s, b, t)
        for b in range(B):
            assert (res[ps][1][b] == res["15"][1][b]).all(), (ps, b)


@pytest.mark.parametrize("flags", [0x700, 0x300, 0x1], ids=["vocals_only", "two_targets", "no_wiener"])
def test_persistent_gemm_with_skipped_targets(pkg, tmp_path, monkeypatch, flags):
    """The persistent walk is flattened over the ACTIVE targets (BASELINE config 1 skips three, umx_hip.h UMX_FLAG_SKIP_TARGET): one,
    two and four targets in the tile list must give the ping-pong kernel's bits (stems of every lane, carried state)."""
    H, N, B = 512, 900 * 1024, 16
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=59), H, compress=False)
    waves = [pkg.ggml.synth_audio(N - 211 * b, 1700 + b) for b in range(B)]
    res = {}
    for ps in ("0", None):
        if ps is None:
            monkeypatch.delenv("UMX_GEMM_PS", raising=False)
        else:
            monkeypatch.setenv("UMX_GEMM_PS", ps)
        eng = pkg.Engine.from_file(path, N, tracks=B, quantised=True)
        outs = eng.infer_batch(waves, flags)
        names = [eng.gemm_kernel_name(m) for m in range(4)]  # (fc1 / fc2 of ONE target have fewer 256 x 256 tiles than CUs: the 128 x 128 kernel)
        assert all(names[m] == ("gemm_planes_ps_kernel" if ps is None else "gemm_planes_pp_kernel") for m in (1, 3)), (ps, names)
        res[ps] = (outs, [eng.track_stream_get(b) for b in range(B)])
        eng.close()
    for b in range(B):
        assert (res[None][1][b] == res["0"][1][b]).all(), b
        for t in range(4):
            assert (res[None][0][b][t] == res["0"][0][b][t]).all(), (b, t)


def test_groups_of_16_lanes_one_launch_after_the_other_give_a_lane_its_bits(pkg, tmp_path, monkeypatch):
    """Contexts that csrc/lstm_batch8.h does not take -- hidden 256 / 128, fp32-resident W_hh, or UMX_LSTM8_MIN_LANES=99 -- run
    lstm_batch_kernel, one group of 16 lanes per launch, the groups of a larger context one after the other (round 6: the side-by-side,
    in-turn and twelve-wave forms of rounds 2-5 are gone).  A lane's stems and carried state must be the bits it has in a context of
    one group, whichever group it sits in: hidden 1024 on the older kernel (20 and 40 lanes, persistent and per-step), hidden 256
    (24 lanes), and fp32-resident weights (20 lanes)."""
    N = 24 * 1024
    for H, quantised, env, cases in ((1024, True, "99", ((20, 17, 0), (40, 35, 0), (40, 9, pkg.FLAG_LSTM_STEPWISE))),
                                     (256, True, None, ((24, 21, 0),)),
                                     (512, False, None, ((20, 18, 0),))):
        if env is None:
            monkeypatch.delenv("UMX_LSTM8_MIN_LANES", raising=False)
        else:
            monkeypatch.setenv("UMX_LSTM8_MIN_LANES", env)
        path = str(tmp_path / f"m{H}{int(quantised)}.bin")
        pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=71), H, compress=False)
        track = [pkg.ggml.synth_audio(N, 2900 + s) for s in range(2)]
        other = [pkg.ggml.synth_audio(N, 2950 + i) for i in range(40)]
        results = []
        for B, lane, flags in ((3, 1, 0),) + cases:
            eng = pkg.Engine.from_file(path, N, tracks=B, quantised=quantised)
            outs = []
            for s in range(2):
                batch = [other[(i + s) % 40] for i in range(B)]
                batch[lane] = track[s]
                outs.append(eng.infer_batch(batch, flags)[lane])
            assert eng.lstm_kernel_name() == "lstm_batch_kernel", (H, B)
            results.append((outs, eng.track_stream_get(lane), B, lane, flags))
            eng.close()
        for outs, state, B, lane, flags in results[1:]:
            assert (state == results[0][1]).all(), (H, B, lane, flags)
            for s in range(2):
                for t in range(4):
                    assert (outs[s][t] == results[0][0][s][t]).all(), (H, B, lane, flags, s, t)


def test_umxhq_width_runs_the_octet_recurrence(pkg, po, tmp_path):
    """hidden 512 (umxhq; src/model.cpp:109-114,136-137 read the width from the file): csrc/lstm_batch8.h with LSTM hidden 256 -- four
    column shards per chain, eight octets side by side, 64 lanes in ONE launch -- against the oracle per lane, and the bits of the same
    track in a small context and through the per-step driver."""
    H, N, B = 512, 24 * 1024, 40
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=73), H, compress=False)
    om = po.Model.load(path)
    waves = [[pkg.ggml.synth_audio(N - 131 * b, 3500 + 10 * b + s) for b in range(B)] for s in range(2)]
    eng = pkg.Engine.from_file(path, N, tracks=B, quantised=True)
    got = [eng.infer_batch(w) for w in waves]
    assert eng.lstm_kernel_name() == "lstm_batch8_kernel" and eng.lstm_was_persistent()
    states = [eng.track_stream_get(b) for b in range(B)]
    eng.close()
    for b in (0, 9, 39):
        ref, ref_state = _oracle_track(po, om, H, [waves[0][b], waves[1][b]], N)
        for s in range(2):
            for t in range(4):
                assert float(np.abs(got[s][b][t] - ref[s][t]).max()) < TOL_WAVE, (b, s, t)
        assert rel_l2(states[b], ref_state) < TOL_STAGE, b
    for B2, flags in ((3, 0), (12, pkg.FLAG_LSTM_STEPWISE)):
        small = pkg.Engine.from_file(path, N, tracks=B2, quantised=True)
        outs = [small.infer_batch([waves[s][9], waves[s][0], waves[s][39]] + [waves[s][1]] * (B2 - 3), flags) for s in range(2)]
        assert small.lstm_kernel_name() == "lstm_batch8_kernel"
        for k, b in enumerate((9, 0, 39)):
            assert (small.track_stream_get(k) == states[b]).all(), (B2, b)
            for s in range(2):
                for t in range(4):
                    assert (outs[s][k][t] == got[s][b][t]).all(), (B2, b, s, t)
        small.close()


def test_two_octets_in_turn_give_the_bits_of_two_launches(pkg, tmp_path, monkeypatch):
    """33 .. 64 lanes, hidden 1024: csrc/lstm_batch8.h with two octets per workgroup IN TURN (one launch per layer; the polls of one
    octet travel under the matrix and gate phases of the other) against the same kernel with one octet per workgroup, the two halves
    of the context one launch after the other (UMX_LSTM8_PAIRED=0), and against the per-step driver: per (unit, lane) the arithmetic
    does not know about turns, so stems and carried state agree bit for bit -- 40 lanes (workgroups with one octet and with two), 64
    lanes, ragged lengths, two segments."""
    H, N = 1024, 40 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=57), H, compress=False)
    for B in (40, 64):
        waves = [[pkg.ggml.synth_audio(N - 97 * b, 3300 + 10 * b + s) for b in range(B)] for s in range(2)]
        res = {}
        for mode in ("halves", "paired", "stepwise"):
            if mode == "halves":
                monkeypatch.setenv("UMX_LSTM8_PAIRED", "0")
            else:
                monkeypatch.delenv("UMX_LSTM8_PAIRED", raising=False)
            eng = pkg.Engine.from_file(path, N, tracks=B, quantised=True)
            outs = [eng.infer_batch(w, pkg.FLAG_LSTM_STEPWISE if mode == "stepwise" else 0) for w in waves]
            assert eng.lstm_kernel_name() == "lstm_batch8_kernel"
            res[mode] = (outs, [eng.track_stream_get(b) for b in range(B)])
            eng.close()
        for mode in ("paired", "stepwise"):
            for s in range(2):
                for b in range(B):
                    for t in range(4):
                        assert (res[mode][0][s][b][t] == res["halves"][0][s][b][t]).all(), (B, mode, s, b, t)
            for b in range(B):
                assert (res[mode][1][b] == res["halves"][1][b]).all(), (B, mode, b)


def test_activation_planes_follow_the_data_range(pkg, po, tmp_path):
    """The plane GEMMs take every activation row as two fp16 planes of the row scaled by a power of two (csrc/gemm_planes.h):
    the scale follows the row, so a near-silent track (1e-5 of full scale, where a fixed-range fp16 split would be all
    subnormals) and an over-driven one (x30) must both stay within the parity tolerance RELATIVE to their own level,
    in the same batch."""
    hidden, N = 128, 24 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(hidden, seed=59), hidden, compress=False)
    om = po.Model.load(path)
    base = pkg.ggml.synth_audio(N, 901)
    gains = [1e-5, 1.0, 30.0]
    eng = pkg.Engine.from_file(path, N, tracks=3)
    got = eng.infer_batch([(base * g).astype(np.float32) for g in gains], pkg.FLAG_DEBUG_TAPS)
    fc1 = [eng.tap(f"fc1#{b}", 1) for b in range(3)]
    eng.close()
    for b, g in enumerate(gains):
        ref, taps = po.umx_inference(om, (base * g).astype(np.float32), n_buf=N, want_taps=True)
        assert rel_l2(fc1[b], taps["fc1_out"][1]) < TOL_STAGE, (g, "fc1")
        for t in range(4):
            scale = max(float(np.abs(ref[t]).max()), 1e-30)
            assert float(np.abs(got[b][t] - ref[t]).max()) / scale < 2e-4, (g, t)


def test_production_configuration_at_full_size_against_the_oracle(pkg, po, tmp_path):
    """What `bench.py` times by default, value-checked: UMX-L width, the full 60 s segment (T = 2584), several track lanes
    in one context -- 256 x 256 plane-GEMM tiles that straddle lanes, the batched matrix-core recurrence, the fused
    Wiener / inverse-STFT kernel -- every stage tap and the stems of every lane against the oracle, two carried
    segments for one lane."""
    hidden, N, B = 1024, pkg.SEGMENT_SAMPLES, 3
    path = str(tmp_path / "m.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(hidden, seed=61), hidden)
    om = po.Model.load(path)
    eng = pkg.Engine.from_file(path, N, tracks=B)
    assert eng.T == 2584 and eng.lstm_is_batched()
    waves = [[pkg.ggml.synth_audio(N if b != 1 else N - 12345, 300 + 10 * b + s) for s in range(2)] for b in range(B)]
    states = [po.stream_state(hidden) for _ in range(B)]
    for s in range(2):
        batch = [waves[b][s] if (s == 0 or b == 0) else None for b in range(B)]  # second call: lane 0 only
        got = eng.infer_batch(batch, pkg.FLAG_DEBUG_TAPS)
        assert eng.lstm_was_persistent()
        for b in range(B):
            if batch[b] is None:
                continue
            ref, taps = po.umx_inference(om, batch[b], n_buf=N, state=states[b], want_taps=True)
            for t in range(4):
                for name, key, tol in (("fc1", "fc1_out", TOL_STAGE), ("lstm", "lstm_out", TOL_STAGE), ("fc2", "fc2_out", TOL_STAGE),
                                       ("mask", "mask", TOL_STAGE), ("target_mag", "target_mag", TOL_STAGE), ("y", "y", 2e-4)):
                    err = rel_l2(eng.tap(f"{name}#{b}", t), taps[key][t])
                    assert err < tol, (s, b, t, name, err)
                    assert err < (REG_Y if name == "y" else REG_STAGE), ("regression bound", s, b, t, name, err)
                assert float(np.abs(got[b][t] - ref[t]).max()) < REG_WAVE, (s, b, t)
            assert rel_l2(eng.track_stream_get(b), states[b]) < REG_STAGE, (s, b)
    eng.close()


@pytest.mark.parametrize("B", [32, 48, 64])
def test_the_bench_configuration_is_value_checked_at_full_size(pkg, po, tmp_path, B):
    """what `bench.py` times by default -- 32 (and 48) track lanes x the full 60 s segment
    (T = 2584) through lstm_batch8_kernel (octets of 8 lanes x column shards of 64 units; more than 32 lanes: two octets per workgroup
    in turn) and plane-GEMM launches of 82,688 (124,032 / 165,376) rows -- against the ORACLE, not only `outputs_finite`.  Two distinct tracks
    duplicated over the lanes: a lane of the first and a lane of the second group (0 and 17; 40 of the third for B = 48)
    against the oracle on every stage tap, the stems and the carried state; every duplicate bitwise equal to its original;
    and bitwise equal to the same track in a 3-lane context (one group: lstm_batch_kernel)."""
    hidden, N = 1024, pkg.SEGMENT_SAMPLES
    path = str(tmp_path / "m.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(hidden, seed=67), hidden)
    om = po.Model.load(path)
    tracks = [pkg.ggml.synth_audio(N, 410), pkg.ggml.synth_audio(N - 4321, 411)]
    which = [(i * 7 + i // 16) % 2 for i in range(B)]  # both tracks in every group of 16 lanes
    which[0], which[17] = 0, 1
    checked = [0, 17] + ([40] if B > 32 else []) + ([55] if B > 48 else [])
    eng = pkg.Engine.from_file(path, N, tracks=B)
    assert eng.T == 2584 and eng.lstm_is_batched()
    got = eng.infer_batch([tracks[which[i]] for i in range(B)], pkg.FLAG_DEBUG_TAPS)
    assert eng.lstm_was_persistent()
    assert eng.lstm_kernel_name() == "lstm_batch8_kernel"  # octets of 8 lanes x column shards of 64 units (more than 32 lanes: two launches per layer)
    refs = []
    for k in range(2):
        st = po.stream_state(hidden)
        ref, taps = po.umx_inference(om, tracks[k], n_buf=N, state=st, want_taps=True)
        refs.append((ref, taps, st))
    for b in checked:
        ref, taps, st = refs[which[b]]
        for t in range(4):
            for name, key, tol in (("fc1", "fc1_out", TOL_STAGE), ("lstm", "lstm_out", TOL_STAGE), ("fc2", "fc2_out", TOL_STAGE),
                                   ("mask", "mask", TOL_STAGE), ("target_mag", "target_mag", TOL_STAGE), ("y", "y", 2e-4)):
                err = rel_l2(eng.tap(f"{name}#{b}", t), taps[key][t])
                assert err < tol, (B, b, t, name, err)
                assert err < (REG_Y if name == "y" else REG_STAGE), ("regression bound", B, b, t, name, err)
            assert float(np.abs(got[b][t] - ref[t]).max()) < REG_WAVE, (B, b, t)
        assert rel_l2(eng.track_stream_get(b), st) < REG_STAGE, (B, b)
    first = {0: which.index(0), 1: which.index(1)}
    states = {k: eng.track_stream_get(first[k]) for k in (0, 1)}
    for b in range(B):
        k = which[b]
        assert (eng.track_stream_get(b) == states[k]).all(), (B, b)
        for t in range(4):
            assert (got[b][t] == got[first[k]][t]).all(), (B, b, t)
    keep = {k: ([got[first[k]][t].copy() for t in range(4)], states[k]) for k in (0, 1)}
    # the instruction stream bench.py times: the same call WITHOUT the debug taps (the recurrence then writes the next GEMM's planes
    # only, no fp32 rows) -- stems and carried state against the oracle, and the bits of the tapped call
    eng.track_stream_reset(-1)
    plain = eng.infer_batch([tracks[which[i]] for i in range(B)])
    for b in checked:
        ref, taps, st = refs[which[b]]
        for t in range(4):
            assert float(np.abs(plain[b][t] - ref[t]).max()) < REG_WAVE, ("no taps", B, b, t)
            assert (plain[b][t] == got[b][t]).all(), ("no taps", B, b, t)
        assert rel_l2(eng.track_stream_get(b), st) < REG_STAGE, ("no taps", B, b)
    del got, plain
    eng.close()
    small = pkg.Engine.from_file(path, N, tracks=3)
    g3 = small.infer_batch([tracks[1], tracks[0], tracks[1]])
    for lane, k in ((0, 1), (1, 0), (2, 1)):
        assert (small.track_stream_get(lane) == keep[k][1]).all(), (B, lane)
        for t in range(4):
            assert (g3[lane][t] == keep[k][0][t]).all(), (B, lane, t)
    small.close()


def test_batched_kernel_agrees_with_single_track_kernel(pkg, tmp_path):
    """Same track through the single-track (VALU) kernel and the batched (matrix-core) kernel: different summation
    order, so not bitwise -- but far inside the parity tolerance."""
    H, N = 1024, 32 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=47), H, compress=False)
    waves = [pkg.ggml.synth_audio(N, 700 + s) for s in range(2)]
    e1 = pkg.Engine.from_file(path, N)
    a = [e1.infer_segment(w, pkg.FLAG_DEBUG_TAPS) for w in waves]
    la = e1.tap("lstm", 3)
    sa = e1.stream_get()
    e1.close()
    e2 = pkg.Engine.from_file(path, N, tracks=2)
    b = [e2.infer_batch([w, w], pkg.FLAG_DEBUG_TAPS) for w in waves]
    lb = e2.tap("lstm#1", 3)
    sb = e2.track_stream_get(1)
    e2.close()
    assert not e1.lstm_is_batched() if e1.h else True
    assert rel_l2(lb, la) < 1e-5
    assert rel_l2(sb, sa) < 1e-5
    for s in range(2):
        for t in range(4):
            assert (b[s][0][t] == b[s][1][t]).all()  # two lanes fed the same audio: same bits
            assert np.abs(b[s][1][t] - a[s][t]).max() < 1e-5


def test_reset_mode_runs_the_segments_of_one_track_as_lanes(pkg, po, model_small):
    """UMX_FLAG_RESET_SEGMENTS (SURVEY 8(e) mode table; a declared deviation from umx.cpp:167-171,226-227): every segment of a track
    from a zero lstm_data, the segments as the track lanes of one call (here 4 lanes: a 6-segment track takes two passes).  Against
    the ORACLE run segment by segment from a zero state (lstm.cpp:86-99 in front of every umx_inference) through the C++ host
    driver's overlap-add, bit for bit against the same engine fed one segment at a time from a reset state, and different from the
    default (carry) mode, which is unchanged."""
    path, om, targets = model_small
    N = 16 * 1024
    wave = pkg.ggml.synth_audio(int(N * 4.3), 77)
    eng = pkg.Engine(targets, 128, N, tracks=4)

    def oracle_seg(w):
        return po.umx_inference(om, w, n_buf=N, state=po.stream_state(128))[0]

    def engine_seg(w):
        eng.track_stream_reset(-1)
        return eng.infer_batch([w])[0]
    for shift in (None, 4033):
        run = (lambda be: pkg.split_inference(be, wave, N)) if shift is None else (lambda be: pkg.shift_inference(be, wave, N, offset=shift))
        ref = run(pkg.make_backend(oracle_seg))
        one = run(pkg.make_backend(engine_seg))
        carry = eng.separate(wave, shift_offset=shift)
        got = eng.separate(wave, flags=pkg.FLAG_RESET_SEGMENTS, shift_offset=shift)
        again = eng.separate(wave, shift_offset=shift)
        for t in range(4):
            assert float(np.abs(got[t] - ref[t]).max()) < TOL_WAVE, (shift, t)
            assert (got[t] == one[t]).all(), (shift, t)
            assert (again[t] == carry[t]).all(), (shift, t)      # the default mode is what it was
        assert max(float(np.abs(got[t] - carry[t]).max()) for t in range(4)) > 0  # ... and is not reset mode
    # one track only, and only on a track-batched context
    with pytest.raises(RuntimeError):
        eng.separate_many([wave, wave], flags=pkg.FLAG_RESET_SEGMENTS)
    eng.close()
    single = pkg.Engine(targets, 128, N)
    with pytest.raises(RuntimeError):
        single.separate(wave, flags=pkg.FLAG_RESET_SEGMENTS)
    single.close()


def test_idle_lane_keeps_its_state_and_lanes_reset_independently(pkg, model_small):
    path, om, targets = model_small
    N = 16 * 1024
    eng = pkg.Engine(targets, 128, N, tracks=3)
    w = [pkg.ggml.synth_audio(N, 600 + i) for i in range(3)]
    eng.infer_batch(w)
    st = [eng.track_stream_get(i) for i in range(3)]
    assert all(np.abs(s).max() > 0 for s in st)
    out = eng.infer_batch([w[1], None, w[0]])  # lane 1 sits out
    assert out[1] is None
    assert (eng.track_stream_get(1) == st[1]).all()
    assert not (eng.track_stream_get(0) == st[0]).all()
    eng.track_stream_reset(2)
    assert np.abs(eng.track_stream_get(2)).max() == 0
    assert (eng.track_stream_get(1) == st[1]).all()
    # set/get round trip, and a lane restarted from a saved state reproduces its continuation
    a = eng.infer_batch([None, w[2], None])[1]
    eng.track_stream_set(1, st[1])
    b = eng.infer_batch([None, w[2], None])[1]
    assert all((a[t] == b[t]).all() for t in range(4))
    # argument errors
    with pytest.raises(RuntimeError):
        eng.infer_batch([w[0]] * 4)  # more lanes than the context has
    with pytest.raises(RuntimeError):
        eng.infer_batch([None, None, None])
    eng.close()
    with pytest.raises(RuntimeError):
        pkg.Engine(targets, 128, N, tracks=pkg.MAX_TRACKS + 1)


def test_whole_tracks_as_track_lanes_and_the_batch_cli(pkg, po, tmp_path):
    """umx_hip_separate_tracks: tracks of different lengths through one context, their segments in lock step (a finished
    track's lane idles): every track equals what it gets alone on a batched-kernel context (bitwise) and the oracle's
    shift_inference (tolerance); then the same through `umx-batch` on wav files."""
    import subprocess
    from pathlib import Path
    H, N = 128, 16 * 1024
    path = str(tmp_path / "m.bin.gz")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=17), H)
    om = po.Model.load(path)
    lens = [int(N * 3.4), N // 2, int(N * 1.9)]
    waves = [pkg.ggml.synth_audio(L, 40 + i) for i, L in enumerate(lens)]
    offs = [4033, None, 700]
    eng = pkg.Engine.from_file(path, N, tracks=3)
    got = eng.separate_many(waves, shift_offsets=offs)
    eng.close()
    one = pkg.Engine.from_file(path, N, lstm_batched=True)
    for i in range(3):
        alone = one.separate(waves[i], shift_offset=offs[i])
        ref = po.shift_inference(om, waves[i], N, offs[i]) if offs[i] is not None else po.split_inference(om, waves[i], N)
        for t in range(4):
            assert (got[i][t] == alone[t]).all(), (i, t)
            assert np.abs(got[i][t] - ref[t]).max() < TOL_WAVE, (i, t)
    one.close()
    # the files form: production segment size, the reference's shipped 5.94 s tracks
    gold = Path(__file__).parent / "golden"
    out = tmp_path / "out"
    cli = Path(pkg.HERE) / "umx-batch"
    r = subprocess.run([str(cli), path, str(out), str(gold / "gspi_stereo.wav"), str(gold / "gspi_mono.wav")], capture_output=True, text=True,
                       env={**__import__("os").environ, "UMX_SHIFT_OFFSET": "4033"}, timeout=600)
    assert r.returncode == 0, r.stderr
    for name in ("gspi_stereo", "gspi_mono"):
        wave, _ = pkg.wav_load(gold / f"{name}.wav")
        ref = po.shift_inference(om, wave, pkg.SEGMENT_SAMPLES, 4033)
        for t in range(4):
            g, ch = pkg.wav_load(out / name / f"target_{t}.wav")
            assert ch == 2 and np.abs(g - ref[t]).max() < TOL_WAVE, (name, t)
    bad = subprocess.run([str(cli), path], capture_output=True, text=True)
    assert bad.returncode == 1 and "Usage" in bad.stderr
    # two inputs with the same file name must not overwrite each other, and every file -- in whatever batch
    # of lanes it lands -- gets the reference's one unseeded rand() % 22050 = 4033 (umx.cpp:115) unless UMX_SHIFT_OFFSET says otherwise
    import shutil
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    shutil.copy(gold / "gspi_stereo.wav", tmp_path / "a" / "x.wav")
    shutil.copy(gold / "gspi_mono.wav", tmp_path / "b" / "x.wav")
    out2 = tmp_path / "out2"
    env2 = {k: v for k, v in __import__("os").environ.items() if k != "UMX_SHIFT_OFFSET"}
    r = subprocess.run([str(cli), path, str(out2), str(tmp_path / "a" / "x.wav"), str(tmp_path / "b" / "x.wav")], capture_output=True, text=True,
                       env=env2, timeout=600)
    assert r.returncode == 0, r.stderr
    for name, src in (("x", "gspi_stereo"), ("x_2", "gspi_mono")):
        for t in range(4):
            g, _ = pkg.wav_load(out2 / name / f"target_{t}.wav")
            ref, _ = pkg.wav_load(out / src / f"target_{t}.wav")  # written above with UMX_SHIFT_OFFSET=4033
            assert (g == ref).all(), (name, t)
