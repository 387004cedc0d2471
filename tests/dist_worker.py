"""Worker for tests/test_distributed_gloo.py: launched by torch.distributed.run with world_size 2,
gloo backend, CPU only.  Exercises the multi-process code paths of umx.cpp_amd/multigpu.py with the
oracle as the segment backend (test infrastructure)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg, po = ge.load_package(), ge.load_oracle()
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    po.set_num_threads(2)
    H, N = 64, 4 * 4096
    om = po.Model.from_arrays(H, pkg.ggml.synth_weights(H, seed=5))
    state = [po.stream_state(H)]

    def seg(w):
        return po.umx_inference(om, w, n_buf=N, state=state[0])[0]

    def reset():
        state[0] = po.stream_state(H)

    wave = pkg.ggml.synth_audio(int(N * 2.6), 12)
    res = mg.separate_track_reset_mode(seg, reset, wave, N, dist=dist, rank=rank, world=world)
    if rank == 0:
        np.save(out_dir / "reset_mode.npy", np.stack(res))
    # track sharding + the bench timing contract
    tracks = mg.shard_tracks(5, rank, world)
    calls = []
    dt = mg.timed_region(lambda: calls.append(1), lambda: None, steps=3, warmup=2, dist=dist, world=world)
    t = torch.tensor([len(tracks), len(calls)], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        np.save(out_dir / "meta.npy", np.array([int(t[0]), int(t[1]), dt > 0]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
