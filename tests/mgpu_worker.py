"""Worker for test_gpu_mgpu.py: launched by torch.distributed.run with world_size W, every rank on cuda:0 (the GPU
box has one device).  The ranks exchange the RCCL rendezvous ids over gloo and then run the C++ multi-GPU track
driver (include/umx_mgpu.h: RCCL send / recv on device pointers); rank 0 saves the stems."""
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
    model_path, out_dir, N, L, seed, off = sys.argv[1], Path(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = ge.load_package()
    eng = pkg.Engine.from_file(model_path, N)
    ids = torch.zeros(pkg.MGPU_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        ids = torch.frombuffer(bytearray(pkg.mgpu_unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(ids, src=0)
    try:
        mg = pkg.MultiGpuTrack(eng, rank, world, bytes(ids.numpy().tobytes()))
    except pkg.UmxError as e:
        if rank == 0:
            (out_dir / "mgpu_error.txt").write_text(str(e))
        dist.barrier()
        return
    wave = pkg.ggml.synth_audio(L, seed)
    res = mg.separate(wave, shift_offset=None if off < 0 else off)
    if rank == 0:
        np.save(out_dir / "mgpu.npy", np.stack(res))
    dist.barrier()
    mg.close()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
