"""Worker for test_gpu_parity.py::test_carry_mode_two_processes_one_gpu: launched by torch.distributed.run
with world_size 2; both ranks drive their own engine on cuda:0 and exchange the per-layer LSTM state over
gloo (the messages are 4 x 4 x hidden/2 floats; on a multi-GPU node the same driver runs with one rank per
GPU and backend "nccl")."""
import os
import sys
from pathlib import Path

import numpy as np
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402


def main():
    model_path, out_dir, N, L, seed = sys.argv[1], Path(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = ge.load_package()
    mg = __import__("importlib").import_module("umx_cpp_amd.multigpu")
    eng = pkg.Engine.from_file(model_path, N)
    wave = pkg.ggml.synth_audio(L, seed)
    res = mg.separate_track_carry_mode(mg.EnginePhases(eng), wave, N, dist=dist, rank=rank, world=world)
    if rank == 0:
        np.save(out_dir / "carry.npy", np.stack(res))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
