"""Co-residency safety of the persistent LSTM launches (they need their whole grid resident):
  * several contexts on one GPU driven from several threads pass one process-wide admission gate -- no timeout, and
    every context gets the bits it gets when it runs alone;
  * if a launch gives up all the same (another process holding the CUs; simulated with UMX_FLAG_DEBUG_LSTM_ABORT, which
    makes layer 1 abort half way exactly like a timed-out poll), umx_hip_sync restores the pre-segment stream state and
    runs the queued segments again with the per-step driver: same bits as an undisturbed run, UMX_OK, and the context
    keeps working."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def umxl(pkg, tmp_path_factory):
    H = 1024
    path = str(tmp_path_factory.mktemp("umxl") / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=91), H, compress=False)
    return path


def _run_segments(pkg, eng, waves, flags=0):
    import torch
    N = waves[0].shape[1]
    ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda() for w in waves]
    outs = [[torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)] for _ in waves]
    torch.cuda.synchronize()
    for i in range(len(waves)):
        eng.infer_segment_device(ins[i].data_ptr(), N, [o.data_ptr() for o in outs[i]], flags)
    eng.sync()
    return [[o.cpu().numpy() for o in seg] for seg in outs], eng.stream_get()


def test_two_umxl_contexts_from_two_threads(pkg, umxl):
    import torch
    torch.zeros(1).cuda()
    N, NSEG = 48 * 1024, 5
    waves = [[pkg.ggml.synth_audio(N, 100 * k + i) for i in range(NSEG)] for k in range(2)]
    alone = []
    for k in range(2):
        eng = pkg.Engine.from_file(umxl, N)
        alone.append(_run_segments(pkg, eng, waves[k]))
        assert eng.lstm_mode() >= 1
        eng.close()
    engs = [pkg.Engine.from_file(umxl, N) for _ in range(2)]
    res, errs = [None, None], []

    def work(k):
        try:
            for _ in range(3):  # several rounds: the two contexts' launches interleave differently every time
                engs[k].stream_reset()
                res[k] = _run_segments(pkg, engs[k], waves[k])
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert engs[k].lstm_mode() >= 1  # still the persistent kernel: nothing timed out
        assert (res[k][1] == alone[k][1]).all()
        for i in range(NSEG):
            for t in range(4):
                assert (res[k][0][i][t] == alone[k][0][i][t]).all(), (k, i, t)
        engs[k].close()


@pytest.mark.parametrize("tracks", [1, 3, 19, 37])  # 19: three octets of lanes (lstm_batch8_kernel, one per workgroup); 37: two octets per workgroup in turn, most second octets empty
def test_timeout_is_recovered_with_the_same_bits(pkg, umxl, tracks):
    import torch
    torch.zeros(1).cuda()
    N, NSEG = 32 * 1024, 3
    waves = [pkg.ggml.synth_audio(N, 300 + i) for i in range(NSEG)]
    eng = pkg.Engine.from_file(umxl, N, tracks=tracks)
    if tracks == 1:
        ref, ref_state = _run_segments(pkg, eng, waves)
        eng.stream_reset()
        got, state = _run_segments(pkg, eng, waves, pkg.FLAG_DEBUG_LSTM_ABORT)
        assert eng.last_error().startswith("recovered")
        assert (state == ref_state).all()
        for i in range(NSEG):
            for t in range(4):
                assert (got[i][t] == ref[i][t]).all(), (i, t)
        # the context keeps working (per-step driver from now on), host-pointer form included
        eng.stream_reset()
        again = [eng.infer_segment(w) for w in waves]
        assert eng.lstm_mode() == 0
        for i in range(NSEG):
            for t in range(4):
                assert (np.ascontiguousarray(again[i][t].T).ravel() == ref[i][t]).all()
    else:
        batch = [waves[(i) % NSEG] for i in range(tracks)]
        ref = [eng.infer_batch(batch), eng.infer_batch(batch[::-1])]
        ref_state = [eng.track_stream_get(i) for i in range(tracks)]
        eng.track_stream_reset(-1)
        got = [eng.infer_batch(batch, pkg.FLAG_DEBUG_LSTM_ABORT), eng.infer_batch(batch[::-1])]  # host form: stems copied again
        assert eng.last_error().startswith("recovered")
        for i in range(tracks):
            assert (eng.track_stream_get(i) == ref_state[i]).all()
            for s in range(2):
                for t in range(4):
                    assert (got[s][i][t] == ref[s][i][t]).all(), (s, i, t)
    eng.close()


def test_whole_track_is_retried_after_a_timeout(pkg, model_small):
    path, om, targets = model_small
    N = 16 * 1024
    eng = pkg.Engine(targets, 128, N)
    wave = pkg.ggml.synth_audio(int(N * 3.3), 930)
    ref = eng.separate(wave, shift_offset=4033)
    got = eng.separate(wave, pkg.FLAG_DEBUG_LSTM_ABORT, shift_offset=4033)
    for t in range(4):
        assert (got[t] == ref[t]).all()
    eng.close()


def _host_async(pkg, eng, waves, flags_first=0):
    """Queue every wave through the host-pointer async form (pinned buffers), one sync at the end."""
    import torch
    N = waves[0].shape[1]
    ins = [torch.from_numpy(np.ascontiguousarray(w.T).ravel()).pin_memory() for w in waves]
    outs = [[torch.empty(2 * N, dtype=torch.float32).pin_memory() for _ in range(4)] for _ in waves]
    for i in range(len(waves)):
        eng.infer_batch_ptrs([ins[i].data_ptr()], [N], [o.data_ptr() for o in outs[i]], flags_first if i == 0 else 0, where="host_async")
    eng.sync()
    return [[o.numpy().copy() for o in seg] for seg in outs], eng.stream_get()


def test_recovery_replays_many_queued_host_calls_on_their_own_audio(pkg, umxl):
    """the host-pointer async form stages audio in two internal buffers; with 5 calls queued before one
    sync, call k's staged audio has been overwritten by call k + 2 when the replay starts.  The replay must upload every
    call's own audio again: same stems and state as an undisturbed run."""
    import torch
    torch.zeros(1).cuda()
    N, NSEG = 24 * 1024, 5
    waves = [pkg.ggml.synth_audio(N, 340 + i) for i in range(NSEG)]
    eng = pkg.Engine.from_file(umxl, N)
    ref, ref_state = _host_async(pkg, eng, waves)
    eng.stream_reset()
    got, state = _host_async(pkg, eng, waves, pkg.FLAG_DEBUG_LSTM_ABORT)
    assert eng.last_error().startswith("recovered")
    assert (state == ref_state).all()
    for i in range(NSEG):
        for t in range(4):
            assert (got[i][t] == ref[i][t]).all(), (i, t)
    eng.close()


def test_state_change_between_queued_calls_is_not_replayed_across(pkg, umxl):
    """infer(A, times out); stream_set(S); infer(B); sync: A is repaired before S is applied, B starts from S."""
    import torch
    torch.zeros(1).cuda()
    N = 24 * 1024
    a, b = pkg.ggml.synth_audio(N, 350), pkg.ggml.synth_audio(N, 351)
    eng = pkg.Engine.from_file(umxl, N)
    ref_a = eng.infer_segment(a)
    S = eng.stream_get()
    eng.stream_set(0.5 * S)
    ref_b = eng.infer_segment(b)
    ref_state = eng.stream_get()
    eng.stream_reset()
    (got_a,), _ = _host_async(pkg, eng, [a], pkg.FLAG_DEBUG_LSTM_ABORT)  # sync inside: recovered here
    ina = torch.from_numpy(np.ascontiguousarray(a.T).ravel()).cuda()
    outs = [torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)]
    eng2 = pkg.Engine.from_file(umxl, N)
    eng2.infer_segment_device(ina.data_ptr(), N, [o.data_ptr() for o in outs], pkg.FLAG_DEBUG_LSTM_ABORT)  # queued, times out
    eng2.stream_set(0.5 * S)  # synchronises through umx_hip_sync: A replayed first, then the state is replaced
    assert eng2.last_error().startswith("recovered")
    for t in range(4):
        assert (outs[t].cpu().numpy() == np.ascontiguousarray(ref_a[t].T).ravel()).all()
    got_b = eng2.infer_segment(b)
    assert (eng2.stream_get() == ref_state).all()
    for t in range(4):
        assert (got_b[t] == ref_b[t]).all()
        assert (got_a[t] == np.ascontiguousarray(ref_a[t].T).ravel()).all()
    eng.close()
    eng2.close()


def test_order_before_gives_up_recovery_and_bounds_the_log(pkg, umxl):
    """A caller that fences with umx_hip_order_before may recycle its buffers: the calls queued so far cannot be replayed,
    so a timeout among them must surface as UMX_ERR_TIMEOUT (not a silent 'recovered' on recycled data)."""
    import torch
    torch.zeros(1).cuda()
    N = 24 * 1024
    w = pkg.ggml.synth_audio(N, 360)
    eng = pkg.Engine.from_file(umxl, N)
    ina = torch.from_numpy(np.ascontiguousarray(w.T).ravel()).cuda()
    outs = [torch.empty(2 * N, dtype=torch.float32, device="cuda") for _ in range(4)]
    eng.infer_segment_device(ina.data_ptr(), N, [o.data_ptr() for o in outs], pkg.FLAG_DEBUG_LSTM_ABORT)
    assert eng.lib.umx_hip_order_before(eng.h, torch.cuda.current_stream().cuda_stream) == 0
    with pytest.raises(pkg.UmxError) as ei:
        eng.sync()
    assert ei.value.code == pkg.ERR_TIMEOUT
    assert np.abs(eng.stream_get()).max() == 0  # documented: state reset to zero
    eng.close()
    # more than 8 calls between syncs: same outcome, and the context keeps working afterwards (per-step driver)
    eng = pkg.Engine.from_file(umxl, N)
    for i in range(10):
        eng.infer_segment_device(ina.data_ptr(), N, [o.data_ptr() for o in outs], pkg.FLAG_DEBUG_LSTM_ABORT if i == 9 else 0)
    with pytest.raises(pkg.UmxError) as ei:
        eng.sync()
    assert ei.value.code == pkg.ERR_TIMEOUT
    got = eng.infer_segment(w)
    assert eng.lstm_mode() == 0 and all(np.isfinite(g).all() for g in got)
    eng.close()
