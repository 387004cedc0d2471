"""Worker for tests/test_distributed_gloo.py::test_target_sharded_track_over_gloo: launched by torch.distributed.run with
world_size 2, 4 or 8 on CPU (gloo).  Drives the C++17 host schedule umx_split_inference_targets (include/umx_host.h) --
one track sharded by source model x segment -- with the ORACLE as the per-target backend (test infrastructure): rank 0
saves the stems, which must equal oracle_split_inference bit for bit."""
import os
import sys
from pathlib import Path

import numpy as np
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


class OracleTargets:
    """umx_target_backend over the oracle: the target network of inference.cpp:70-186 per target (oracle_target_network),
    wiener + istft for the finish.  A layer's outgoing (h, c) depends only on the incoming states of layers <= l, so each
    phase re-runs the target with the states known so far (the last run, in masks(), has them all)."""

    def __init__(self, po, om, H, N):
        self.po, self.om, self.H, self.N, self.Hl = po, om, H, N, H // 2
        self.T = po.nb_frames(N)
        self.carry = {}  # like the engine: a target's state stays where the previous segment left it unless set_state replaces it

    def target_layer_floats(self):
        return 4 * self.Hl

    def mag_floats(self):
        return 2 * self.T * 2049

    def begin(self, chunk, mask):
        self.chunk, self.targets = chunk, [t for t in range(4) if (mask >> t) & 1]
        _, taps = self.po.umx_inference(self.om, chunk, n_buf=self.N, want_taps=True)  # front: spec, |X|, x
        self.spec, self.mix_mag, self.x = taps["spec"], taps["mix_mag"], taps["x"]
        self.inc = {t: self.carry.get(t, np.zeros((3, 2, 2, self.Hl), np.float32)).copy() for t in self.targets}
        self.outg, self.mags = {}, {}

    def set_state(self, l, t, a):
        self.inc[t][l] = np.asarray(a, np.float32).reshape(2, 2, self.Hl)

    def _run(self, t):
        st = self.inc[t].copy().reshape(-1)
        r = self.po.target_network(self.om, t, self.x, self.mix_mag, st)
        return r, st.reshape(3, 2, 2, self.Hl)

    def layer(self, l):
        for t in self.targets:
            _, st = self._run(t)
            self.outg[(l, t)] = st[l].copy().reshape(-1)

    def get_state(self, l, t):
        return self.outg[(l, t)]

    def masks(self):
        for t in self.targets:
            r, st = self._run(t)
            self.mags[t], self.carry[t] = r["target_mag"], st

    def get_mag(self, t):
        return self.mags[t].reshape(-1)

    def set_mag(self, t, a):
        self.mags[t] = np.asarray(a, np.float32).reshape(2, self.T, 2049)

    def finish(self):
        ys = self.po.wiener(self.spec, [self.mags[t] for t in range(4)])
        n = self.chunk.shape[1]
        return [self.po.istft(y, n, self.N) for y in ys]

    def discard(self):
        pass


def main():
    out_dir = Path(sys.argv[1])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg, po = ge.load_package(), ge.load_oracle()
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    po.set_num_threads(1)
    H, N = 64, 4 * 4096
    om = po.Model.from_arrays(H, pkg.ggml.synth_weights(H, seed=5))
    wave = pkg.ggml.synth_audio(int(N * 4.1), 14)
    res = mg.separate_track_target_mode(OracleTargets(po, om, H, N), wave, N, dist=dist, rank=rank, world=world)
    if rank == 0:
        np.save(out_dir / f"targets_w{world}.npy", np.stack(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
