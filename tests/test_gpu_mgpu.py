"""The C++17 multi-GPU track driver (include/umx_mgpu.h, host/mgpu.cpp): BASELINE config 4.  On the one-GPU test box:
  * world 1 through the device-resident path (same kernels, no communicator) must equal the single-GPU whole-track
    driver bit for bit;
  * world 1 LOOPBACK: a one-rank RCCL communicator, every LSTM-state hop, every target-magnitude exchange and every stem
    gather through a grouped RCCL self send + receive on the streams a real run uses, the outgoing buffers poisoned in
    between -- the RCCL path executes on one GPU, segment-sharded and target-sharded, bitwise equal to the one-GPU driver;
    a persistent-LSTM timeout inside it is found through the status all-reduce and the track retried;
  * world 2 / 3 with all ranks on the one device where the RCCL build accepts that (otherwise skipped with the reason)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_world1_device_path_equals_whole_track_driver(pkg, tmp_path):
    H, N = 1024, 24 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=71), H, compress=False)
    eng = pkg.Engine.from_file(path, N)
    mg = pkg.MultiGpuTrack(eng)
    for L, off in ((int(N * 3.3), None), (int(N * 2.1), 4033), (N // 2, 20000)):
        wave = pkg.ggml.synth_audio(L, 800 + L % 7)
        ref = eng.separate(wave, shift_offset=off)
        got = mg.separate(wave, shift_offset=off)
        for t in range(4):
            assert (got[t] == ref[t]).all(), (L, off, t)
    mg.close()
    eng.close()


@pytest.mark.parametrize("by_target", [False, True], ids=["segments", "targets"])
def test_rccl_loopback_executes_every_hop_and_equals_the_one_gpu_driver(pkg, tmp_path, by_target):
    H, N = 1024, 24 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=73), H, compress=False)
    eng = pkg.Engine.from_file(path, N)
    mg = pkg.MultiGpuTrack(eng, loopback=True, by_target=by_target)
    for L, off, flags in ((int(N * 3.3), 4033, 0), (int(N * 2.1), None, pkg.FLAG_NO_WIENER), (int(N * 1.6), 20000, 0x700)):
        wave = pkg.ggml.synth_audio(L, 820 + L % 7)
        ref = eng.separate(wave, flags=flags, shift_offset=off)
        got = mg.separate(wave, shift_offset=off, flags=flags)
        st = mg.stats()
        nseg = len(pkg.segment_plan(L + (0 if off is None else max(22050 - off, off)), N)[0])
        assert st["state_hops"] == 2 * 4 * 3 * (nseg - 1) and st["stem_transfers"] == 4 * nseg and st["magnitude_transfers"] == 8 * nseg
        assert st["retries"] == 0
        for t in range(4):
            assert np.isfinite(got[t]).all()
            assert (got[t] == ref[t]).all(), (L, off, t)
    # a persistent launch that gives up half way (as if another process held the CUs): every rank learns it from the
    # status all-reduce, nothing is handed out, the track is run again (this rank on the per-step driver): same bits
    wave = pkg.ggml.synth_audio(int(N * 2.4), 830)
    ref = eng.separate(wave, shift_offset=4033)
    got = mg.separate(wave, shift_offset=4033, flags=pkg.FLAG_DEBUG_LSTM_ABORT)
    assert mg.stats()["retries"] == 1
    for t in range(4):
        assert (got[t] == ref[t]).all(), t
    mg.close()
    eng.close()


def test_watchdog_ends_a_wait_for_a_peer_that_never_sends(pkg, tmp_path, monkeypatch):
    """host/mgpu.cpp: every host-side wait of a track polls its streams against a deadline (UMX_MGPU_TIMEOUT_MS) and asks the
    communicators for asynchronous errors.  In loopback the engine's stream is held in front of one state hop for twice the
    deadline (UMX_MGPU_DEBUG_STALL: a sleeping host function; an unmatched receive is refused by RCCL's group call outright): to
    this rank that is what a peer that never posts its side of a transfer looks like.  The call must come back with the
    watchdog's error after the deadline (not after the stall, not never), the driver must refuse further tracks, destroy must
    return, and a fresh driver on the same engine must work again, bit for bit."""
    import time
    H, N = 1024, 24 * 1024
    path = str(tmp_path / "m.bin")
    pkg.ggml.write_model(path, pkg.ggml.synth_weights(H, seed=79), H, compress=False)
    eng = pkg.Engine.from_file(path, N)
    wave = pkg.ggml.synth_audio(int(N * 2.6), 840)
    ref = eng.separate(wave, shift_offset=4033)
    monkeypatch.setenv("UMX_MGPU_TIMEOUT_MS", "2000")
    monkeypatch.setenv("UMX_MGPU_DEBUG_STALL", "1")
    mg = pkg.MultiGpuTrack(eng, loopback=True)
    t0 = time.time()
    with pytest.raises(pkg.UmxError) as ei:
        mg.separate(wave, shift_offset=4033)
    assert 1.9 < time.time() - t0 < 30.0
    assert "watchdog" in str(ei.value)
    with pytest.raises(pkg.UmxError):  # the communicators were aborted: the driver is dead, not half alive
        mg.separate(wave, shift_offset=4033)
    mg.close()
    monkeypatch.delenv("UMX_MGPU_DEBUG_STALL")
    mg = pkg.MultiGpuTrack(eng, loopback=True)
    got = mg.separate(wave, shift_offset=4033)
    for t in range(4):
        assert (got[t] == ref[t]).all(), t
    mg.close()
    eng.close()


def test_mgpu_refuses_a_track_batched_context(pkg, model_small):
    path, om, targets = model_small
    eng = pkg.Engine(targets, 128, 16 * 1024, tracks=2)
    with pytest.raises(pkg.UmxError):
        pkg.MultiGpuTrack(eng)
    eng.close()


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_path_ranks_on_one_device(pkg, model_small, tmp_path, world):
    path, om, targets = model_small
    N, L, seed = 16 * 1024, int(16 * 1024 * 4.6), 811
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), str(ROOT / "tests" / "mgpu_worker.py"), path, str(tmp_path), str(N), str(L), str(seed), "4033"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    if (tmp_path / "mgpu_error.txt").exists():
        pytest.skip("RCCL refused several ranks on one device: " + (tmp_path / "mgpu_error.txt").read_text())
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "mgpu.npy")
    eng = pkg.Engine(targets, 128, N)
    one = np.stack(eng.separate(pkg.ggml.synth_audio(L, seed), shift_offset=4033))
    eng.close()
    assert got.shape == one.shape
    assert (got == one).all()


@pytest.mark.parametrize("mode", ["track", "targets"])
def test_bench_track_modes_run_through_rccl_in_loopback(mode):
    """`bench.py --mode track | targets` (BASELINE config 4) is the driver's entry for N > 1; on the one-GPU box it runs through
    the same C++ driver with every transfer an RCCL self send + receive (--loopback): one JSON line with the contract's
    fields, the RCCL operation counts of the track, finite outputs."""
    import json
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--mode", mode, "--loopback", "--track-seconds", "130", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["outputs_finite"] and d["scaling"] == "strong"
    nseg = d["config"]["segments"]
    rc = d["config"]["rccl"]
    assert rc["state_hops"] == 2 * 4 * 3 * (nseg - 1) and rc["stem_transfers"] == 4 * nseg and rc["retries"] == 0
