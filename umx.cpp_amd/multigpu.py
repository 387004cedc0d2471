"""Multi-GPU sharding of the separation path, one process per GPU over torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

The unit of work is a segment (umx.cpp:214-227).  What is exact and what is not:

* `shard_tracks`      -- independent TRACKS round-robin over ranks.  Exact (tracks share nothing); no
                         data-path collective.  This is what `bench.py --gpus N` measures (weak scaling).
* `separate_track_reset_mode` -- ONE track, its segments round-robin over ranks, every segment started
                         from a zero LSTM state.  The reference carries the state across segments
                         (SURVEY F3), so this DEVIATES from it; the overlap-add itself is exact (each
                         output sample has at most two contributors and a + b == b + a in IEEE).  Rank 0
                         gathers the weighted stems with point-to-point transfers and normalises.
* exact single-track splitting needs the per-layer state hand-off (wavefront of SURVEY 8e); inside one
  GPU that wavefront is what csrc/engine.hip's two pipeline slots already do; across GPUs it is the next
  step (`umx_hip_stream_get/set` expose the 32 KB state).

The functions only need a `segment_fn((2,n) float32 array) -> 4 x (2,n)` and are therefore testable on
CPU with any backend.
"""
import numpy as np


def shard_tracks(n_tracks, rank, world):
    """Indices of the tracks this rank separates."""
    return list(range(rank, n_tracks, world))


def _weights(chunk_len, segment_samples):
    # umx.cpp:197-206, 246: triangular transition weight of sample k, weight[k % chunk_len]
    k = np.arange(chunk_len) % chunk_len
    N = segment_samples
    raw = np.where(k < N // 2, k + 1, N - k).astype(np.float32)
    return (raw / np.float32(N // 2)).astype(np.float32)


def separate_track_reset_mode(segment_fn, reset_fn, wave, segment_samples, dist=None, rank=0, world=1, device="cpu"):
    """Reset-mode split of one track over `world` ranks; returns 4 x (2,L) on rank 0, None elsewhere."""
    import torch
    wave = np.asarray(wave, np.float32)
    L = wave.shape[1]
    N = segment_samples
    stride = int((1 - 0.25) * N)  # umx.cpp:181
    offsets = list(range(0, L, stride))
    mine = [i for i in range(len(offsets)) if i % world == rank]
    local = {}
    for i in mine:
        off = offsets[i]
        n = min(N, L - off)
        if reset_fn is not None:
            reset_fn()  # reset mode: every segment starts from zero state
        stems = segment_fn(np.ascontiguousarray(wave[:, off:off + n]))
        w = _weights(n, N)
        local[i] = np.stack([np.asarray(s, np.float32) * w[None, :] for s in stems])  # (4,2,n), pre-weighted
    if world > 1:
        # point-to-point gather of the weighted stems to rank 0 (<= 85 MB per segment at full size)
        if rank == 0:
            for i in range(len(offsets)):
                if i % world != 0:
                    n = min(N, L - offsets[i])
                    buf = torch.empty((4, 2, n), dtype=torch.float32, device=device)
                    dist.recv(buf, src=i % world)
                    local[i] = buf.cpu().numpy()
        else:
            for i in mine:
                dist.send(torch.from_numpy(local[i]).to(device), dst=0)
            return None
    out = np.zeros((4, 2, L), np.float32)
    sum_w = np.zeros(L, np.float32)
    for i, off in enumerate(offsets):  # segment order = the reference's accumulation order
        n = min(N, L - off)
        out[:, :, off:off + n] += local[i]
        sum_w[off:off + n] += _weights(n, N)
    out /= sum_w[None, None, :]
    return [out[t] for t in range(4)]


def timed_region(step_fn, sync_fn, steps, warmup, dist=None, world=1, device=None):
    """bench.py's timing contract: W untimed warm-up steps, barrier + sync on both sides of exactly
    K timed steps, MAX over ranks.  Returns seconds."""
    import time
    import torch
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if world > 1:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if world > 1:
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt
