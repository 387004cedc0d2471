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


class OraclePhases:
    """Phased backend over the oracle (test infrastructure): a layer's outgoing (h, c) depends only on the
    incoming states of layers <= l, so each phase re-runs the whole segment with the states known so far."""

    def __init__(self, po, om, H, N):
        self.po, self.om, self.H, self.N = po, om, H, N
        self.Hl = H // 2

    def layer_floats(self):
        return 4 * 4 * self.Hl

    def _full(self):  # [target][layer][dir][h|c][Hl]
        st = self.po.stream_state(self.H).reshape(4, 3, 2, 2, self.Hl)
        for l, a in self.inc.items():
            st[:, l] = a.reshape(4, 2, 2, self.Hl)
        return st

    def begin(self, chunk):
        self.chunk, self.inc, self.outg = chunk, {}, {}

    def set_layer(self, l, a):
        self.inc[l] = np.array(a, np.float32)

    def layer(self, l):
        st = self._full()
        flat = st.reshape(-1)
        self.stems = self.po.umx_inference(self.om, self.chunk, n_buf=self.N, state=flat)[0]
        self.outg[l] = flat.reshape(4, 3, 2, 2, self.Hl)[:, l].copy().reshape(-1)

    def get_layer(self, l):
        return self.outg[l]

    def end(self):
        return self.stems  # the layer-2 run had every incoming state in place


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
    # exact carry mode: segments alternate between the ranks, per-layer state travels point to point
    wave2 = pkg.ggml.synth_audio(int(N * 3.3), 13)
    res = mg.separate_track_carry_mode(OraclePhases(po, om, H, N), wave2, N, dist=dist, rank=rank, world=world)
    if rank == 0:
        np.save(out_dir / "carry_mode.npy", np.stack(res))
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
