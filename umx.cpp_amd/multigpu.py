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
* `separate_track_carry_mode` -- ONE track, its segments round-robin over ranks, EXACT: the reference
                         carries every chain's (h, c) from one segment into the next, and layer l
                         of segment s needs nothing else of segment s-1.  Rank s % world runs segment s phase
                         by phase (umx_hip_segment_begin / _lstm_layer / _end); before layer l it receives
                         that layer's 32 KB state from the rank that ran segment s-1 and afterwards sends its
                         own on: a wavefront over GPUs, point-to-point messages only (no collective has a
                         role here).  The front (STFT, fc1, W_ih) and back (fc2, fc3, Wiener, iSTFT) stages
                         of different segments overlap freely.  Bit-identical to the single-GPU result.
  Inside one GPU the same wavefront is what csrc/engine.hip's two pipeline slots do.

The drivers only need a `segment_fn((2,n) float32 array) -> 4 x (2,n)` or a phased backend (see
`EnginePhases`) and are therefore testable on CPU with any stand-in backend.
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
    return _gather_overlap_add(local, mine, offsets, L, N, dist, rank, world, device)


def _gather_overlap_add(local, mine, offsets, L, N, dist, rank, world, device):
    """Weighted stems of every segment -> rank 0 (point-to-point), overlap-add in segment order, / sum_w."""
    import torch
    if world > 1:
        # <= 85 MB per segment at full size
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


class EnginePhases:
    """Phased backend over the C-ABI (include/umx_hip.h: umx_hip_segment_begin ... umx_hip_stream_set_layer)."""

    def __init__(self, engine, flags=0):
        self.e, self.flags = engine, flags

    def layer_floats(self):
        return int(self.e.lib.umx_hip_stream_layer_floats(self.e.h))

    def begin(self, chunk):
        self.e.segment_begin(chunk, self.flags)

    def layer(self, l):
        self.e.segment_lstm_layer(l)

    def end(self):
        return self.e.segment_end()

    def get_layer(self, l):
        return self.e.stream_get_layer(l)

    def set_layer(self, l, a):
        self.e.stream_set_layer(l, a)


def separate_track_carry_mode(backend, wave, segment_samples, dist=None, rank=0, world=1, device="cpu"):
    """Exact split of one track over `world` ranks (state-carry wavefront); 4 x (2,L) on rank 0, else None.
    `backend`: begin(chunk) / layer(l) / end() -> 4 x (2,n) / get_layer(l) / set_layer(l, a) / layer_floats().

    A ctypes view of the C++17 host driver `umx_split_inference_carry` (include/umx_host.h, host/split.cpp): the
    schedule, the weighting and the overlap-add live there; this function only adapts the backend object and a
    torch.distributed point-to-point transport (gloo in the CPU tests) to its callback tables.  The device-resident
    form of the same schedule -- RCCL send / recv on device pointers, no host bounce -- is host/mgpu.cpp."""
    import ctypes as C
    import torch
    from . import (P2P, P2P_FN, PH_BEGIN_FN, PH_END_FN, PH_LAYER_FN, PH_STATE_FN, PhasedBackend, HostError, host_lib, _fp)
    wave = np.asarray(wave, np.float32)
    L = wave.shape[1]
    nf = backend.layer_floats()
    keep, failure = [], []

    def guard(fn):
        def wrapped(*a):
            try:
                r = fn(*a)
                return 0 if r is None else r
            except Exception as e:  # noqa: BLE001 - surfaced through the driver's error code
                failure.append(e)
                return 13
        return wrapped

    def _begin(_u, audio, n):
        backend.begin(np.ascontiguousarray(np.ctypeslib.as_array(audio, shape=(n, 2)).T))
        keep[:] = [n]

    def _layer(_u, l):
        backend.layer(l)

    def _end(_u, out):
        stems = backend.end()
        n = keep[0]
        for t in range(4):
            np.ctypeslib.as_array(out[t], shape=(n, 2))[:, :] = np.asarray(stems[t], np.float32).T

    def _get(_u, l, st):
        np.ctypeslib.as_array(st, shape=(nf,))[:] = np.asarray(backend.get_layer(l), np.float32).ravel()

    def _set(_u, l, st):
        backend.set_layer(l, np.ctypeslib.as_array(st, shape=(nf,)).copy())

    pending = []

    def _send(_u, buf, n, dst):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(n,)).copy()).to(device)
        pending.append((dist.isend(t, dst=dst), t))  # buffered: the driver may reuse `buf` at once

    def _recv(_u, buf, n, src):
        t = torch.empty(n, dtype=torch.float32, device=device)
        dist.recv(t, src=src)
        np.ctypeslib.as_array(buf, shape=(n,))[:] = t.cpu().numpy()

    cbs = (PH_BEGIN_FN(guard(_begin)), PH_LAYER_FN(guard(_layer)), PH_END_FN(guard(_end)), PH_STATE_FN(guard(_get)),
           PH_STATE_FN(guard(_set)), P2P_FN(guard(_send)), P2P_FN(guard(_recv)))
    be = PhasedBackend(cbs[0], cbs[1], cbs[2], cbs[3], cbs[4], nf, None)
    p2p = P2P(cbs[5], cbs[6], None)
    a = np.ascontiguousarray(wave.T).ravel()
    outs = [np.empty(2 * L, np.float32) for _ in range(4)] if rank == 0 else None
    arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs]) if rank == 0 else None
    err = C.create_string_buffer(256)
    rc = host_lib().umx_split_inference_carry(C.byref(be), C.byref(p2p) if world > 1 else None, rank, world,
                                              a.ctypes.data_as(_fp), L, segment_samples, arr, err)
    for req, _t in pending:
        req.wait()
    if rc:
        if failure:
            raise failure[0]
        raise HostError(rc, err.value.decode())
    return [np.ascontiguousarray(o.reshape(L, 2).T) for o in outs] if rank == 0 else None


def _torch_p2p(dist, device, guard):
    """umx_p2p over torch.distributed: buffered sends (isend of a copy), blocking receives."""
    import torch
    from . import P2P, P2P_FN
    pending = []

    def _send(_u, buf, n, dst):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(n,)).copy()).to(device)
        pending.append((dist.isend(t, dst=dst), t))  # buffered: the driver may reuse `buf` at once

    def _recv(_u, buf, n, src):
        t = torch.empty(n, dtype=torch.float32, device=device)
        dist.recv(t, src=src)
        np.ctypeslib.as_array(buf, shape=(n,))[:] = t.cpu().numpy()
    cbs = (P2P_FN(guard(_send)), P2P_FN(guard(_recv)))
    return P2P(cbs[0], cbs[1], None), cbs, pending


def separate_track_target_mode(backend, wave, segment_samples, dist=None, rank=0, world=1, device="cpu"):
    """Exact split of one track over `world` ranks by SOURCE MODEL x segment (include/umx_host.h:
    umx_split_inference_targets; world = gcd(world, 4) target groups x pipeline stages); 4 x (2,L) on rank 0, else None.
    `backend`: begin(chunk, target_mask) / layer(l) / get_state(l, t) / set_state(l, t, a) / masks() / get_mag(t) /
    set_mag(t, a) / finish() -> 4 x (2,n) / discard() / target_layer_floats() / mag_floats().  The schedule lives in the
    C++17 host driver; this function adapts the backend object and a torch.distributed transport to its callback tables.
    The device-resident form (RCCL on device pointers) is host/mgpu.cpp with UMX_MGPU_BY_TARGET."""
    import ctypes as C
    from . import (PH_END_FN, PH_LAYER_FN, TG_BEGIN_FN, TG_MAG_FN, TG_STATE_FN, TG_VOID_FN, TargetBackend, HostError, host_lib, _fp)
    wave = np.asarray(wave, np.float32)
    L = wave.shape[1]
    nf, nm = backend.target_layer_floats(), backend.mag_floats()
    keep, failure = [], []

    def guard(fn):
        def wrapped(*a):
            try:
                r = fn(*a)
                return 0 if r is None else r
            except Exception as e:  # noqa: BLE001 - surfaced through the driver's error code
                failure.append(e)
                return 13
        return wrapped

    def _begin(_u, audio, n, mask):
        backend.begin(np.ascontiguousarray(np.ctypeslib.as_array(audio, shape=(n, 2)).T), mask)
        keep[:] = [n]

    def _finish(_u, out):
        stems = backend.finish()
        for t in range(4):
            np.ctypeslib.as_array(out[t], shape=(keep[0], 2))[:, :] = np.asarray(stems[t], np.float32).T

    def _get_state(_u, l, t, st):
        np.ctypeslib.as_array(st, shape=(nf,))[:] = np.asarray(backend.get_state(l, t), np.float32).ravel()

    def _get_mag(_u, t, m):
        np.ctypeslib.as_array(m, shape=(nm,))[:] = np.asarray(backend.get_mag(t), np.float32).ravel()
    cbs = (TG_BEGIN_FN(guard(_begin)), PH_LAYER_FN(guard(lambda _u, l: backend.layer(l))), TG_STATE_FN(guard(_get_state)),
           TG_STATE_FN(guard(lambda _u, l, t, st: backend.set_state(l, t, np.ctypeslib.as_array(st, shape=(nf,)).copy()))),
           TG_VOID_FN(guard(lambda _u: backend.masks())), TG_MAG_FN(guard(_get_mag)),
           TG_MAG_FN(guard(lambda _u, t, m: backend.set_mag(t, np.ctypeslib.as_array(m, shape=(nm,)).copy()))),
           PH_END_FN(guard(_finish)), TG_VOID_FN(guard(lambda _u: backend.discard())))
    be = TargetBackend(*cbs, nf, nm, None)
    p2p, p2p_keep, pending = _torch_p2p(dist, device, guard) if world > 1 else (None, None, [])
    a = np.ascontiguousarray(wave.T).ravel()
    outs = [np.empty(2 * L, np.float32) for _ in range(4)] if rank == 0 else None
    arr = (_fp * 4)(*[o.ctypes.data_as(_fp) for o in outs]) if rank == 0 else None
    err = C.create_string_buffer(256)
    rc = host_lib().umx_split_inference_targets(C.byref(be), C.byref(p2p) if world > 1 else None, rank, world,
                                                a.ctypes.data_as(_fp), L, segment_samples, arr, err)
    for req, _t in pending:
        req.wait()
    if rc:
        if failure:
            raise failure[0]
        raise HostError(rc, err.value.decode())
    return [np.ascontiguousarray(o.reshape(L, 2).T) for o in outs] if rank == 0 else None


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
