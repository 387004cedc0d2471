"""world_size-2 gloo (CPU) coverage of the N > 1 paths: track sharding, the bench timing
contract (barrier + MAX over ranks), and the reset-mode single-track split with its P2P overlap-add
gather, compared with a single-process evaluation of the same mode and with the exact (carry) result."""
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def test_two_rank_gloo_paths(pkg, po, tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "tests" / "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    meta = np.load(tmp_path / "meta.npy")
    assert meta[0] == 5          # 5 tracks were split 3 + 2 over the ranks
    assert meta[1] == 2 * 5      # every rank ran warmup 2 + steps 3
    assert meta[2] == 1
    two_rank = np.load(tmp_path / "reset_mode.npy")

    # single-process evaluation of the same reset-mode algorithm
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    H, N = 64, 4 * 4096
    om = po.Model.from_arrays(H, pkg.ggml.synth_weights(H, seed=5))
    state = [po.stream_state(H)]

    def seg(w):
        return po.umx_inference(om, w, n_buf=N, state=state[0])[0]

    def reset():
        state[0] = po.stream_state(H)
    wave = pkg.ggml.synth_audio(int(N * 2.6), 12)
    one = np.stack(mg.separate_track_reset_mode(seg, reset, wave, N))
    assert (one == two_rank).all()  # the sharding changes nothing bit-wise

    # against the reference semantics (state carried, umx.cpp:167-171): segment 0 is identical, later
    # segments differ -- the deviation reset mode must declare
    exact = np.stack(po.split_inference(om, wave, N))
    first = int(0.75 * N)
    assert np.abs(exact[:, :, :first] - one[:, :, :first]).max() < 1e-6
    assert np.abs(exact[:, :, first:] - one[:, :, first:]).max() > 1e-6

    # carry mode (exact): two ranks, per-layer (h, c) handed from segment to segment, must give
    # the reference's split_inference -- the overlap-add arithmetic is identical, so bit for bit
    wave2 = pkg.ggml.synth_audio(int(N * 3.3), 13)
    carry = np.load(tmp_path / "carry_mode.npy")
    exact2 = np.stack(po.split_inference(om, wave2, N))
    assert carry.shape == exact2.shape
    assert np.abs(carry - exact2).max() <= 1e-6 * max(1.0, np.abs(exact2).max())


import pytest


@pytest.mark.parametrize("world", [4, 8, 2])
def test_target_sharded_track_over_gloo(pkg, po, tmp_path, world):
    """north_star: "the four source models ... shard naturally".  umx_split_inference_targets (host/split.cpp) over gloo:
    world 4 = four target groups, world 8 = four groups x a two-stage segment pipeline (LSTM state between the stages of a
    group, target magnitudes to the rank that filters the segment, stems to rank 0), world 2 = two groups of two targets.
    The oracle is the per-target backend; the result must be the oracle's split_inference bit for bit."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "target_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env={**__import__("os").environ, "OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / f"targets_w{world}.npy")
    H, N = 64, 4 * 4096
    om = po.Model.from_arrays(H, pkg.ggml.synth_weights(H, seed=5))
    wave = pkg.ggml.synth_audio(int(N * 4.1), 14)
    exact = np.stack(po.split_inference(om, wave, N))
    assert got.shape == exact.shape
    assert (got == exact).all(), float(np.abs(got - exact).max())


def test_shard_plan_covers_every_target_and_segment_once():
    """host/shard_plan.h restated: every (target, segment) has exactly one owner, every segment one filtering rank among
    its owners, adjacent ring edges never share a colour."""
    def gcd4(w):
        return 4 if w % 4 == 0 else 2 if w % 2 == 0 else 1
    for world in (1, 2, 3, 4, 6, 8):
        G = gcd4(world)
        P = world // G
        for s in range(10):
            owners = {}
            for r in range(world):
                if s % P == r // G:
                    for t in range(4):
                        if t % G == r % G:
                            assert t not in owners
                            owners[t] = r
            assert sorted(owners) == [0, 1, 2, 3]
            wr = (s // P) % G + G * (s % P)
            assert wr in owners.values()
        col = [2 if (P % 2 == 1 and p == P - 1 and P > 1) else p & 1 for p in range(P)]
        if P > 1:
            assert all(col[p] != col[(p + 1) % P] for p in range(P)) or P == 2 and col[0] != col[1]


def test_shard_tracks():
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    assert mg.shard_tracks(10, 3, 8) == [3]
    assert mg.shard_tracks(10, 1, 8) == [1, 9]
    assert sorted(sum((mg.shard_tracks(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_python_overlap_add_weights_equal_the_host_driver(pkg):
    """multigpu._weights (used by the carry / reset mode gather) must be the host driver's umx_transition_weight
    (umx.cpp:197-206) bit for bit, or the multi-GPU result could not equal the single-GPU one."""
    import ctypes as C
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    lib = pkg.host_lib()
    lib.umx_transition_weight.restype = C.c_float
    lib.umx_transition_weight.argtypes = [C.c_int, C.c_int, C.c_int]
    for N, n in ((16384, 16384), (16384, 5000), (2646000, 2646000), (2646000, 123457)):
        w = mg._weights(n, N)
        ks = [k for k in list(range(0, n, max(1, n // 997))) + [n - 1, N // 2 - 1, N // 2] if 0 <= k < n]
        for k in ks:
            assert w[k] == np.float32(lib.umx_transition_weight(k, n, N)), (N, n, k)


def test_survivor_of_a_dead_rank_returns_an_error(tmp_path):
    """A rank that dies mid-track (host/split.cpp's carry schedule over gloo): the survivor's next receive fails, the transport
    callback reports it, and umx_split_inference_carry returns an error code within the transport's timeout instead of hanging
    (host/mgpu.cpp does the same for the device path: ncclCommAbort + return).  The ranks are started by hand: torchrun would
    kill the survivor itself."""
    import os
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = {**os.environ, "RANK": str(rank), "WORLD_SIZE": "2", "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
               "MASTER_PORT": str(port), "OMP_NUM_THREADS": "1"}
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "kill_worker.py"), str(tmp_path)], env=env, cwd=str(ROOT),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        rcs = [p.wait(timeout=180) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs[1] == 7, (rcs, procs[1].stderr.read()[-2000:])  # rank 1 died where it was told to
    assert rcs[0] == 0, (rcs, procs[0].stderr.read()[-2000:])
    verdict, seconds = (tmp_path / "rank0.txt").read_text().splitlines()[:2]
    assert verdict.startswith("error"), verdict  # not "completed", and not a hang (the wait above would have timed out)
    assert float(seconds) < 120, seconds
    assert not (tmp_path / "rank1.txt").exists()
