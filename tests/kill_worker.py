"""Worker for tests/test_distributed_gloo.py::test_survivor_of_a_dead_rank_returns_an_error: two ranks over gloo (started WITHOUT
torchrun, which would tear the survivor down itself) drive the C++17 host schedule umx_split_inference_carry (host/split.cpp) with a
trivial phased backend.  Rank 1 dies in the middle of the track (os._exit inside its second segment's layer 1); rank 0 must come
back from the driver with an error -- a failed receive surfaces as the driver's status code (umx.cpp has no such path: its only
failure mode is exit(1)) -- instead of waiting for state that will never arrive.  Mirrors host/mgpu.cpp's abort on a local error."""
import datetime
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


class FakePhases:
    """begin / layer / end with a 16-float state per layer; stem t = (t + 1) * chunk + mean(state)."""

    def __init__(self, die_at=None):
        self.state = np.zeros((3, 16), np.float32)
        self.calls, self.die_at = 0, die_at

    def layer_floats(self):
        return 16

    def begin(self, chunk):
        self.chunk = chunk

    def layer(self, l):
        self.calls += 1
        if self.die_at is not None and self.calls == self.die_at:
            os._exit(7)  # the process is gone: no exception, no goodbye to the peer
        self.state[l] = self.state[l] * 0.5 + float(self.chunk.mean()) + l

    def get_layer(self, l):
        return self.state[l]

    def set_layer(self, l, a):
        self.state[l] = a

    def end(self):
        return [(t + 1) * self.chunk + self.state.mean() for t in range(4)]


def main():
    out_dir = Path(sys.argv[1])
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=30))
    rank, world = dist.get_rank(), dist.get_world_size()
    ge.load_package()
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    N = 4096
    wave = np.random.default_rng(3).uniform(-1, 1, (2, int(N * 6.2))).astype(np.float32)
    # rank 1 runs the odd segments: its second segment's layer 1 is its 5th layer call
    backend = FakePhases(die_at=5 if rank == 1 else None)
    t0 = time.time()
    try:
        mg.separate_track_carry_mode(backend, wave, N, dist=dist, rank=rank, world=world)
        verdict = "completed"
    except Exception as e:  # noqa: BLE001 - any error is the expected outcome on the survivor
        verdict = f"error {type(e).__name__}: {str(e)[:200]}"
    (out_dir / f"rank{rank}.txt").write_text(f"{verdict}\n{time.time() - t0:.1f}\n")
    os._exit(0)  # do not wait for a clean shutdown of a group whose peer is dead


if __name__ == "__main__":
    main()
