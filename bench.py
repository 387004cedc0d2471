#!/usr/bin/env python
"""bench.py -- realtime factor of UMX-L 4-stem separation on 60 s segments (BASELINE.json metric).

A "step" = one pass of the hot path (umx_inference, inference.cpp:12-207) over one synthetic 60 s stereo segment of
EVERY track lane of the context: STFT -> 4 x [fc1/bn/tanh -> 3-layer BiLSTM -> fc2 -> fc3 -> mask] -> Wiener EM ->
4 x iSTFT.  Consecutive steps are consecutive segments of the same tracks: the streaming LSTM state carries over
(umx.cpp:167-171).  The default workload is BASELINE config 3 (4 stems + Wiener, the full umx_inference) on
--tracks independent tracks per GPU (default 64: their LSTM recurrences share one matrix-core launch per layer,
SURVEY 8f-4; ~180 GB of the 288 GB of HBM); --tracks 1 is the single-track, latency-optimised engine; --no-wiener gives config 2, --vocals-only
config 1.

What the JSON line holds (one line, rank 0):
  value            audio-seconds per wall-second, whole job, inputs and stems RESIDENT IN HBM (the contract's number);
                   the workload is --tracks independent tracks per GPU, one 60 s segment of each per step
  value_pcie       the same K steps with pinned HOST buffers in and out: H2D audio + all kernels + D2H stems inside the
                   timed region (SURVEY 8(d)'s unit of work), transfers overlapped through the two pipeline slots
  value_single_segment         BASELINE config 3 as it is worded -- ONE track, its 60 s segments one after the other
                   through the single-track engine (two segments in flight: the exact wavefront), HBM resident;
  value_single_segment_pcie    ... the same with pinned host buffers; lone_segment_ms = one segment with nothing else in flight
                               (lone_segment_ms_latency_context: the same in a one-track context created with UMX_CREATE_GEMM_PLANES)
  checked_max_abs  after the timed region: two lanes of the batched engine, from a reset state, against the single-track
                   engine on the same audio (max |difference| over the stems; `outputs_checked` = below 1e-5).  The
                   oracle comparison of this configuration is tests/test_gpu_batch.py (full size, 32, 48 and 64 lanes).
  single_track     details of the --tracks 1 engine on the same GPU
  roofline         the dominant kernel by device time: live HIP-event duration per launch (events on the engine's own
                   streams); `achieved` / `frac` = ALGORITHMIC flops per launch / duration / peak of the
                   pipe the kernel issues on; frac_issued = the same on issued matrix products; frac_latency for the
                   recurrence; `traffic` = HBM bytes per launch from the rocprofv3 PMC summary named in `traffic_source`
                   (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md), null if it is absent
  kernels          the same figures for every kernel family, so each fraction can be recomputed from the line
  cpu_baseline     the oracle (reference flag set, compiled on this box) on this box's host cores on the FULL 60 s segment:
                   config 3 on all cores (the headline leg) + `legs` for config 1 / config 3 on one thread and on all cores

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank separates its own independent
tracks (weak scaling; the path shards by track, no data-path collective); barrier + synchronize on both sides,
MAX over ranks, rank 0 prints ONE JSON line.  --mode track instead shards ONE track's segments over the ranks
(BASELINE config 4, exact state carry; see umx.cpp_amd/multigpu.py).
"""
import argparse
import csv
import glob
import json
import os
import re
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

SEG = 60 * 44100
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
NB, KXA, NOUT = 2049, 2974, 4098


def algorithmic_work(T, H):
    """Algorithmic flops / bytes of ONE launch of each kernel family for one track lane."""
    gemm = {
        "fc1": 2.0 * T * KXA * H * 4,
        "lstm_ih": 2.0 * T * H * (4 * H) * 4,          # one layer, both directions, 4 targets
        "fc2": 2.0 * T * (2 * H) * H * 4,
        "fc3_mask": 2.0 * T * H * NOUT * 4,
    }
    rec = 2.0 * T * (H // 2) * (2 * H) * 2 * 4         # one layer: W_hh.h, 2 dirs, 4 targets
    byt = {
        "stft": 4.0 * (2 * T * 1024 + 2 * T * NB * 2 + 2 * T * NB + T * 2976),
        "wiener": 4.0 * (4 * (2 * T * NB * 3) + (2 * T * NB * 2 + 4 * 2 * T * NB) + 4 * 2 * T * NB * 2),
        "mixphase": 4.0 * (2 * T * NB * 2 + 4 * 2 * T * NB + 4 * 2 * T * NB * 2),
        "istft": 4.0 * (4 * 2 * T * NB * 2 + 4 * T * 4096 * 2),
        "ola": 4.0 * (4 * T * 4096 * 2 + 4 * 2 * T * 1024),
        # round-2 fused forms: all-source statistics (mixture + 4 magnitudes read once); gains + filter + inverse STFT frame
        "wiener_stats4": 4.0 * (2 * T * NB * 2 + 4 * 2 * T * NB),
        # mixture + masks in, the STEMS out (the kernel overlap-adds: the frames no longer reach HBM)
        "wiener_istft": 4.0 * (2 * T * NB * 2 + 4 * 2 * T * NB + 4 * 2 * T * 1024),
    }
    return gemm, rec, byt


def cpu_leg(pkg, weights_path, seconds_audio, threads, flags, label):
    """The oracle built with the reference's Release flags (-O3 -march=native -ffast-math -fopenmp, per-timestep GEMV
    LSTM) on the first `seconds_audio` of the synthetic track as one segment; NOT the Eigen binary."""
    po = ge.load_oracle()
    po.set_num_threads(threads, fast=True)
    om = po.Model.load(weights_path, fast=True)
    n = int(seconds_audio * 44100)
    wave = pkg.ggml.synth_audio(n, seed=0)
    t0 = time.time()
    po.umx_inference(om, wave, flags=flags)
    dt = time.time() - t0
    return {"config": label, "value": round(seconds_audio / dt, 4), "unit": "x realtime (audio-sec / wall-sec)", "cores": threads,
            "kind": "port", "sample": f"first {seconds_audio:g} s of the synthetic track as one segment ({n // 1024 + 1} frames), "
                                      f"{dt:.1f} s wall"}


def cpu_baseline(pkg, weights_path, seconds_all, seconds_one):
    ncores = os.cpu_count()
    built = ge.load_oracle().build_fast_native()  # -march=native of THIS box, not of the build container
    legs = [cpu_leg(pkg, weights_path, seconds_all, ncores, 0, "config 3: 4 stems + Wiener"),
            cpu_leg(pkg, weights_path, seconds_all, ncores, 0x700, "config 1: vocals model only (targets 0-2 skipped)"),
            cpu_leg(pkg, weights_path, seconds_one, 1, 0, "config 3: 4 stems + Wiener"),
            cpu_leg(pkg, weights_path, seconds_one, 1, 0x700, "config 1: vocals model only (targets 0-2 skipped)")]
    head = dict(legs[0])
    head["sample"] += ("; Eigen-equivalent restatement (oracle/, -O3 -march=native -ffast-math -fopenmp, per-timestep GEMV LSTM), "
                       "not the Eigen binary; library " + built + f"; {ncores} OpenMP threads of which the LSTM recurrence can use 8 "
                       "(targets x directions -- the reference's own loop is serial there), the GEMMs all")
    head["legs"] = legs
    return head


def read_traffic(path):
    """profiles/*pmc_fetch_write_per_kernel.csv -> {kernel name: HBM bytes per launch} (2 x FETCH + WRITE, raw KB)."""
    out = {}
    try:
        for r in csv.reader(open(path)):
            if len(r) >= 4 and r[0] != "kernel":
                out[r[0]] = (2.0 * float(r[2]) + float(r[3])) * 1024.0
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--tracks", type=int, default=64,
                    help="independent tracks per GPU run together (track lanes, 1..64): one step = one 60 s segment of EVERY "
                         "track; > 1 selects the batched matrix-core LSTM kernel (SURVEY 8f-4): workgroups of 8 lanes x 64 hidden "
                         "units (lstm_batch8_kernel), > 32 two such octets per workgroup in turn")
    ap.add_argument("--batched-lstm", action="store_true", help="use the batched LSTM kernel also with --tracks 1")
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--segment-samples", type=int, default=SEG)
    ap.add_argument("--no-wiener", action="store_true", help="BASELINE config 2 (mixture phase, no Wiener EM)")
    ap.add_argument("--vocals-only", action="store_true", help="BASELINE config 1 on the GPU (targets 0-2 skipped)")
    ap.add_argument("--stepwise-lstm", action="store_true")
    ap.add_argument("--safe-lstm", action="store_true", help="persistent LSTM kernel without the intra-XCD hand-off")
    ap.add_argument("--lstm-profile", action="store_true", help="print per-phase shader-clock counters of the LSTM kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-seconds", type=float, default=60.0, help="audio seconds of the all-cores CPU legs (60 = the full segment)")
    ap.add_argument("--cpu-sample-seconds-1t", type=float, default=60.0, help="audio seconds of the one-thread CPU legs")
    ap.add_argument("--serial", action="store_true", help="sync after every step (no cross-segment pipelining)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the value_pcie leg")
    ap.add_argument("--no-single-track", action="store_true", help="skip the single-track (tracks = 1) leg")
    ap.add_argument("--track-seconds", type=float, default=600.0,
                    help="also time a whole track of this length through umx_hip_shift_inference (host buffers in and "
                         "out, PCIe included; BASELINE config 4 on one GPU) and report it as 'track'")
    ap.add_argument("--mode", choices=["segments", "track", "targets"], default="segments",
                    help="track: ONE 600 s track, its segments sharded over the ranks with exact LSTM state carry (config 4); "
                         "targets: the same track sharded by source model x segment (gcd(N, 4) target groups x pipeline stages)")
    ap.add_argument("--loopback", action="store_true", help="--mode track / targets on ONE GPU: every transfer through an RCCL self send + receive")
    ap.add_argument("--gemm", choices=["planes", "bf16x3"], default=None,
                    help="dense-stack GEMM flavour: planes (default; fp16 matrix cores, pre-split operands, fp32-class accuracy) or "
                         "bf16x3 (three bf16 terms, operands split while staged)")
    ap.add_argument("--expanded-weights", action="store_true",
                    help="expand the u8/u16 weights at load time (fp32 / three bf16 planes in HBM) instead of keeping "
                         "them quantised in HBM with dequantisation inside the kernels (the default, BASELINE config 5)")
    ap.add_argument("--u8-dequant", action="store_true", help="u8 weights dequantised per element (UMX_CREATE_U8_DEQUANT)")
    ap.add_argument("--traffic-csv", default=None, help="rocprofv3 PMC summary to take roofline.traffic from "
                                                        "(default: the newest profiles/r*_pmc_fetch_write_per_kernel.csv)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = ge.load_package()
    import importlib
    mg = importlib.import_module("umx_cpp_amd.multigpu")
    H, N, B = args.hidden, args.segment_samples, args.tracks
    # synthetic UMX-L-shaped weights in the reference's file format (u8/u16 + scale/offset); every
    # rank writes its own copy (seeded, identical) to avoid a file-system race
    tmpdir = tempfile.mkdtemp(prefix=f"umx_bench_r{rank}_")
    wpath = os.path.join(tmpdir, "ggml-model-synth-u8.bin")
    pkg.ggml.write_model(wpath, pkg.ggml.synth_weights(H, seed=0), H, compress=False)
    if args.mode in ("track", "targets"):
        return bench_track_mode(args, pkg, mg, dist, world, rank, local_rank, dev, wpath)

    def make_engine(tracks, batched):
        return pkg.Engine.from_file(wpath, segment_samples=N, device=local_rank, quantised_resident=not args.expanded_weights,
                                    gemm=args.gemm, tracks=tracks, lstm_batched=batched, u8_dequant=args.u8_dequant)
    eng = make_engine(B, args.batched_lstm)
    T = eng.T
    flags = (pkg.FLAG_NO_WIENER if args.no_wiener else 0) | (pkg.FLAG_LSTM_STEPWISE if args.stepwise_lstm else 0) | \
        (pkg.FLAG_LSTM_FORCE_SAFE if args.safe_lstm else 0) | (pkg.FLAG_LSTM_PROFILE if args.lstm_profile else 0) | \
        (0x700 if args.vocals_only else 0)

    # each rank and each track lane: its own track; (2,n) interleaved, in HBM
    waves = [np.ascontiguousarray(pkg.ggml.synth_audio(N, seed=rank * 16 + b).T).ravel() for b in range(B)]
    audios = [torch.from_numpy(w).to(dev) for w in waves]
    # two output sets: consecutive steps are in flight together (two pipeline slots) and must not share stems
    depth = eng.pipeline_depth()  # that many consecutive steps are in flight together and must not share stems
    out_sets = [[torch.empty(2 * N, dtype=torch.float32, device=dev) for _ in range(4 * B)] for _ in range(depth)]
    ptr_sets = [[o.data_ptr() for o in st_] for st_ in out_sets]
    aptrs = [a.data_ptr() for a in audios]
    nstep = [0]

    def step():
        eng.infer_batch_ptrs(aptrs, [N] * B, ptr_sets[nstep[0] % depth], flags)
        nstep[0] += 1
        if args.serial:
            eng.sync()

    def fence():
        eng.sync()
        torch.cuda.synchronize()

    dd = dist if world > 1 else None
    # W untimed warm-up steps, barrier + synchronize on both sides of exactly K steps, MAX over ranks
    dt = mg.timed_region(step, fence, args.steps, args.warmup, dist=dd, world=world, device=dev)
    # per-stage device time from hipEvents on the engine's own streams.  (a) the last two steps of the timed region
    # (one per pipeline slot; consecutive steps overlap there, so a span includes the other slot's kernels -- this is
    # the duration rocprofv3 reports for the same command); (b) extra steps run one at a time afterwards (each kernel
    # alone on the chip).
    prof_pipelined = eng.lstm_profile() if args.lstm_profile else None
    st = [d for d in (eng.stage_times(slot=i) for i in range(depth)) if d]
    stage_ms = {k: sum(d[k] for d in st) / len(st) for k in st[0]} if st else {}
    kern_ms = {}
    ks = [d for d in (eng.stage_kernel_times(slot=i) for i in range(depth)) if d]
    if ks:
        kern_ms = {k: sum(d[k] for d in ks) / len(ks) for k in ks[0]}
    # the same K steps once more, one at a time (a sync after each): min / median show the box's noise beside the mean of the
    # timed region; the stages of the last one = each kernel alone on the chip
    serial = []
    stage_alone_ms, kern_alone_ms = {}, {}
    for _ in range(max(2, args.steps)):
        t1 = time.perf_counter()
        step()
        eng.sync()
        serial.append((time.perf_counter() - t1) * 1e3)
    stage_alone_ms = eng.stage_times()
    kern_alone_ms = eng.stage_kernel_times()
    finite = bool(all(torch.isfinite(o).all().item() for st_ in out_sets for o in st_))
    lstm_mode, batched = eng.lstm_mode(), eng.lstm_is_batched()
    lstm_kernel = eng.lstm_kernel_name()  # what the engine launched, not a guess from the lane count
    gemm_kernels = [eng.gemm_kernel_name(m) for m in range(4)]  # per stage: fc1, W_ih, fc2, fc3

    # ---- value_pcie: pinned host buffers in and out, the same number of steps timed the same way
    dt_pcie = None
    if not args.no_pcie and world == 1:  # an N = 1 figure, like cpu_baseline (and an exception on one rank must not leave the others at a barrier)
        try:
            h_in = [torch.from_numpy(w).pin_memory() for w in waves]
            h_out = [[torch.empty(2 * N, dtype=torch.float32).pin_memory() for _ in range(4 * B)] for _ in range(depth)]
            hp_in, hp_out = [t.data_ptr() for t in h_in], [[t.data_ptr() for t in s] for s in h_out]
            k = [0]

            def step_pcie():
                eng.infer_batch_ptrs(hp_in, [N] * B, hp_out[k[0] % depth], flags, where="host_async")
                k[0] += 1
            dt_pcie = mg.timed_region(step_pcie, fence, args.steps, max(2, args.warmup // 2), dist=dd, world=world, device=dev)
            finite = finite and bool(all(torch.isfinite(o).all().item() for s in h_out for o in s))
            del h_in, h_out
        except Exception as e:  # noqa: BLE001 - reported, never a reason to lose the main number
            dt_pcie = repr(e)
    # ---- value check of what was timed: two lanes from a reset state, compared below with the single-track
    # engine on the same audio
    check_lanes = sorted({0, B - 1})
    check_stems = None
    if rank == 0 and world == 1 and B > 1 and not args.no_single_track:
        eng.track_stream_reset(-1)
        eng.infer_batch_ptrs(aptrs, [N] * B, ptr_sets[0], flags)
        eng.sync()
        check_stems = {b: [out_sets[0][4 * b + t].cpu().numpy() for t in range(4)] for b in check_lanes}
    weight_bytes = eng.weight_bytes()
    if args.lstm_profile and rank == 0:
        for tag, pr in (("pipelined", prof_pipelined), ("alone", eng.lstm_profile())):
            for layer in range(3):
                for w in range(2):
                    c = pr[layer, w]
                    n = max(int(c[4]), 1)
                    print(f"# lstm {tag} layer {layer} wave {w}: cycles/step poll {c[0] / n:.0f} dot {c[1] / n:.0f} barrier {c[2] / n:.0f} "
                          f"gates {c[3] / n:.0f} failed-polls/step {c[5] / n:.2f} (steps {int(c[4])}) "
                          f"[poll: sleep {c[6] / n:.0f} first-loads {c[7] / n:.0f}]", file=sys.stderr)
        if not batched:
            from collections import Counter
            pl = eng.lstm_placement(8 * (H // 2 // 16))
            cus = Counter((x, (hw >> 8) & 0xff) for x, _, _, hw in pl)  # (XCC, SE | SH | CU of HW_ID)
            print(f"# lstm placement: {len(pl)} workgroups on {len(cus)} distinct CUs; CUs holding 2 or more: "
                  f"{sum(1 for v in cus.values() if v > 1)}; per XCC: {sorted(Counter(x for x, _, _, _ in pl).items())}", file=sys.stderr)
            # timeline of one workgroup (chain 0, slice 5), 64 steps: per wave the stamps loop top, poll ok, dot end, barrier exit, end
            nraw = 960 + 64 * 8 * 5
            buf = (pkg.C.c_ulonglong * nraw)()
            eng._check(eng.lib.umx_hip_debug_lstm_placement(eng.h, buf, nraw))
            tr = np.array(buf[960:], dtype=np.int64).reshape(64, 8, 5)
            if tr.any():
                base = tr[:, 0, 3]  # wave 0's barrier exit of the step = start of its gate phase
                rel = tr[1:] - base[:-1, None, None]  # step s+1's stamps relative to the barrier exit that ended step s
                med = np.median(rel, axis=0)
                print("# lstm timeline (cycles after the previous barrier exit of wave 0; median of 63 steps): wave: loop-top poll-ok dot-end barrier-exit end", file=sys.stderr)
                for w in range(8):
                    print(f"#   wave {w}: " + " ".join(f"{v:6.0f}" for v in med[w]), file=sys.stderr)
                print(f"#   step period (barrier exit to barrier exit, wave 0): median {np.median(np.diff(base)):.0f} "
                      f"min {np.diff(base).min()} max {np.diff(base).max()}", file=sys.stderr)
    eng.close()
    del eng, audios, out_sets
    torch.cuda.empty_cache()

    # ---- single-track engine on the same GPU (latency view), rank 0 of a 1-GPU run only
    single = None
    if rank == 0 and world == 1 and B > 1 and not args.no_single_track:
        e1 = make_engine(1, False)
        a1 = torch.from_numpy(waves[0]).to(dev)
        d1n = e1.pipeline_depth()
        o1 = [[torch.empty(2 * N, dtype=torch.float32, device=dev) for _ in range(4)] for _ in range(d1n)]
        p1 = [[o.data_ptr() for o in s] for s in o1]
        k1 = [0]

        def step1():
            e1.infer_segment_device(a1.data_ptr(), N, p1[k1[0] % d1n], flags)
            k1[0] += 1

        def fence1():
            e1.sync()
            torch.cuda.synchronize()
        d1 = mg.timed_region(step1, fence1, 16, 4)
        s1 = [d for d in (e1.stage_times(slot=i) for i in range(d1n)) if d]
        lstm1 = sum(sum(d[f"lstm_rec{l}"] for l in range(3)) for d in s1) / (3 * len(s1))
        step1()
        e1.sync()
        alone1 = e1.stage_times()
        # one segment with nothing else in flight (sync after each)
        lone = []
        for _ in range(6):
            t1 = time.perf_counter()
            step1()
            e1.sync()
            lone.append((time.perf_counter() - t1) * 1e3)
        # the same 16 back-to-back segments with pinned host buffers in and out
        d1p = None
        try:
            h1 = torch.from_numpy(waves[0]).pin_memory()
            ho1 = [[torch.empty(2 * N, dtype=torch.float32).pin_memory() for _ in range(4)] for _ in range(d1n)]
            kp = [0]

            def step1p():
                e1.infer_batch_ptrs([h1.data_ptr()], [N], [o.data_ptr() for o in ho1[kp[0] % d1n]], flags, where="host_async")
                kp[0] += 1
            d1p = mg.timed_region(step1p, fence1, 16, 4)
        except Exception as e:  # noqa: BLE001
            d1p = repr(e)
        # value check: the batched engine's lanes against this engine, both from a reset state on the same audio
        checked = None
        if check_stems:
            checked = 0.0
            for b in check_lanes:
                e1.stream_reset()
                ab = torch.from_numpy(waves[b]).to(dev)
                e1.infer_segment_device(ab.data_ptr(), N, p1[0], flags)
                e1.sync()
                for t in range(4):
                    checked = max(checked, float(np.abs(o1[0][t].cpu().numpy() - check_stems[b][t]).max()))
        single = {"value": round(16 * (N / 44100.0) / d1, 2), "unit": "x realtime", "ms_per_step": round(d1 / 16 * 1e3, 3),
                  "lone_segment_ms": round(float(np.median(lone)), 3),
                  "value_pcie": round(16 * (N / 44100.0) / d1p, 2) if isinstance(d1p, float) else None,
                  "ms_per_step_pcie": round(d1p / 16 * 1e3, 3) if isinstance(d1p, float) else d1p,
                  "tracks_per_gpu": 1, "lstm_kernel": "single-track (VALU, lstm_persistent_kernel)",
                  "lstm_launch_ms": round(lstm1, 4), "lstm_launch_ms_alone": round(sum(alone1[f"lstm_rec{l}"] for l in range(3)) / 3, 4),
                  "checked_max_abs": checked, "checked_lanes": check_lanes if check_stems else None}
        e1.close()
        # a lone segment in a one-track context created for LATENCY (UMX_CREATE_GEMM_PLANES: the plane GEMMs instead of the staged
        # ones -- their workgroups do not fit beside two recurrence grids, so segments back to back are slower there, DESIGN 4.3;
        # the flavour is a property of the context, not of the load: a segment's bits never depend on what else is in flight)
        if args.gemm is None:
            try:
                e1p = pkg.Engine.from_file(wpath, segment_samples=N, device=local_rank, quantised_resident=not args.expanded_weights,
                                           gemm="planes", tracks=1, lstm_batched=False, u8_dequant=args.u8_dequant)
                lone_p = []
                for i in range(8):
                    t1 = time.perf_counter()
                    e1p.infer_segment_device(a1.data_ptr(), N, p1[0], flags)
                    e1p.sync()
                    if i >= 2:
                        lone_p.append((time.perf_counter() - t1) * 1e3)
                single["lone_segment_ms_latency_context"] = round(float(np.median(lone_p)), 3)
                e1p.close()
            except Exception as e:  # noqa: BLE001
                single["lone_segment_ms_latency_context"] = repr(e)

    if rank == 0:
        seg_sec = N / 44100.0
        value = world * B * args.steps * seg_sec / dt
        gemm, rec, byt = algorithmic_work(T, H)
        flavour = args.gemm or os.environ.get("UMX_GEMM") or ("planes" if batched else "bf16x3")
        def pmc_order(path):
            # profiles/r<round>_v<version>_pmc_fetch_write_per_kernel.csv: newest = largest (round, version) as NUMBERS
            # (a string sort puts r03_v9 behind r03_v10)
            mm = re.match(r"r(\d+)_v(\d+)_", os.path.basename(path))
            return (int(mm.group(1)), int(mm.group(2))) if mm else (-1, -1)
        traffic_src = args.traffic_csv or max(glob.glob(str(ROOT / "profiles" / "r*_pmc_fetch_write_per_kernel.csv")), key=pmc_order, default=None)
        traffic = read_traffic(traffic_src) if traffic_src else {}
        traffic_missing = []

        def find_traffic(*needles):
            # the counter row of THIS kernel or nothing: no fall-back to another kernel's row (a needle that is not in the
            # summary is reported in roofline.traffic_missing, and the entry's traffic is null)
            for kk, v in traffic.items():
                if all(nd in kk for nd in needles):
                    return v
            if traffic:
                traffic_missing.append(" ".join(needles))
            return None

        # ---- one entry per kernel family: launches per step, live duration per launch, work per launch, roof
        def gemm_entry(stage_keys, work_key, products, name, tneedles):
            # the plane GEMMs cover all track lanes in one launch (the stage also holds the small split_planes launch)
            launches = len(stage_keys) * (1 if flavour == "planes" else B)
            # launch_ms = the GEMM kernel itself (event between the stage's split kernel and the GEMM ... next stage); stage_ms = with the split kernel
            stg = sum(stage_ms.get(kk, 0.0) for kk in stage_keys) / launches
            stg_alone = sum(stage_alone_ms.get(kk, 0.0) for kk in stage_keys) / launches
            ms = sum(kern_ms.get(kk, stage_ms.get(kk, 0.0)) for kk in stage_keys) / launches
            ms_alone = sum(kern_alone_ms.get(kk, stage_alone_ms.get(kk, 0.0)) for kk in stage_keys) / launches
            alg = gemm[work_key] * (B if flavour == "planes" else 1)
            issued = alg * products if flavour != "f32" else alg
            peak = BF16_MFMA_PEAK_TF if flavour != "f32" else F32_MFMA_PEAK_TF
            ach = alg / (ms * 1e-3) / 1e12 if ms > 0 else 0.0  # ALGORITHMIC flops / live launch duration
            ach_issued = issued / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"kernel": name, "bound": "mfma", "pipe": "fp16 MFMA" if flavour != "f32" else "fp32 MFMA", "launches_per_step": launches,
                    "launch_ms": round(ms, 4), "launch_ms_alone": round(ms_alone, 4), "kernel_ms": round(ms, 4), "stage_ms": round(stg, 4),
                    "stage_ms_alone": round(stg_alone, 4), "algorithmic_flops_per_launch": alg,
                    "issued_flops_per_launch": issued, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "frac_issued": round(ach_issued / peak, 4),
                    "frac_alone": round(alg / (ms_alone * 1e-3) / 1e12 / peak, 4) if ms_alone > 0 else None,
                    "frac_issued_alone": round(issued / (ms_alone * 1e-3) / 1e12 / peak, 4) if ms_alone > 0 else None,
                    "algorithmic_TFLOPs_alone": round(alg / (ms_alone * 1e-3) / 1e12, 1) if ms_alone > 0 else None,
                    "traffic": find_traffic(*tneedles)}
        planes = flavour in ("planes", "bf16x3")  # both run on the bf16 matrix cores
        exact = planes and not args.expanded_weights and not args.u8_dequant  # integer weights as exact bf16 terms
        # products per fp32 product: planes = 2 fp16 planes per activation x 1 weight plane (u8), or 3 of the 4 products with the 2 (u16 / fp32);
        # bf16x3 = 3 bf16 terms per activation x 1 exact plane (u8) or the 6-product rule
        p8 = (2 if exact else 4) if flavour == "planes" else (3 if exact else 6)
        p16 = 3 if flavour == "planes" else 6  # (a2 x the low weight plane, 2^-22 of the sum, is not formed: csrc/gemm_planes.h)
        # (the PMC summary holds demangled names: "void umx::gemm_planes_ps_kernel<1, 1>(umx::GemmPArgs, int)" = <MODE, planes of B>);
        # which kernel served a stage is what the engine reports (persistent walk gemm_planes_ps_kernel for launches with more 256 x 256
        # tiles than CUs, else gemm_planes_pp_kernel / gemm_planes_kernel; gemm_bf16x3_kernel in one-track contexts)
        kernels = [gemm_entry(["fc1"], "fc1", p8, f"{gemm_kernels[0]}<G_FC1>", (gemm_kernels[0] + "<0,",)),
                   gemm_entry(["lstm_ih0", "lstm_ih1", "lstm_ih2"], "lstm_ih", p8, f"{gemm_kernels[1]}<G_IH>", (gemm_kernels[1] + "<1,",)),
                   gemm_entry(["fc2"], "fc2", p16, f"{gemm_kernels[2]}<G_FC2>", (gemm_kernels[2] + "<2,",)),
                   gemm_entry(["fc3_mask"], "fc3_mask", p16, f"{gemm_kernels[3]}<G_FC3>", (gemm_kernels[3] + "<3,",))]
        lstm_keys = [f"lstm_rec{l}" for l in range(3)]
        lms = sum(stage_ms.get(kk, 0.0) for kk in lstm_keys) / 3
        lms_alone = sum(stage_alone_ms.get(kk, 0.0) for kk in lstm_keys) / 3
        lstm_alg = rec * B
        if batched:
            lp = 2 if (not args.expanded_weights and not args.u8_dequant) else 6  # u8 W_hh: 1 fp16 plane x 2 fp16 planes of h
            groups = (B + 15) // 16  # the matrix instruction is 16 tracks wide: one MFMA phase per group of 16 lanes
            lstm_issued = rec * 16 * groups * lp * (1.25 if lp == 2 else 1.0)  # + the all-ones tile of the u8 form
            if lstm_kernel == "lstm_batch8_kernel":
                # csrc/lstm_batch8.h: N = 16 = an octet of 8 lanes x the 2 planes of h; 34 matrix instructions per wave and step of which 2
                # are the all-ones tile's
                lstm_issued = rec * 8 * ((B + 7) // 8) * 2 * (34.0 / 32.0)
            lname = lstm_kernel
        else:
            lstm_issued = lstm_alg
            lname = lstm_kernel
        lach = lstm_alg / (lms * 1e-3) / 1e12 if lms > 0 else 0.0
        # `achieved` / `frac`: ALGORITHMIC flops per launch / live launch duration / peak of the pipe the kernel issues on
        # (the batched kernels: fp16 matrix cores; the single-track kernel: fp32 VALU = the fp32 roof).  `frac_issued`
        # prices the issued products (2 planes of h + the all-ones row-sum tile) the same way; the algorithmic rate
        # against the fp32 roof stays as `frac_of_fp32_roof_algorithmic` (round 1's figure: 0.097).
        lpeak = BF16_MFMA_PEAK_TF if batched else F32_MFMA_PEAK_TF
        lrate_issued = lstm_issued / (lms * 1e-3) / 1e12 if lms > 0 else 0.0
        kernels.append({"kernel": lname, "bound": "latency", "pipe": "fp16 MFMA" if batched else "fp32 VALU", "launches_per_step": 3,
                        "launch_ms": round(lms, 4),
                        "launch_ms_alone": round(lms_alone, 4), "algorithmic_flops_per_launch": lstm_alg,
                        "issued_flops_per_launch": lstm_issued, "achieved": round(lach, 2), "peak": lpeak,
                        "unit": "TFLOP/s", "frac": round(lach / lpeak, 4), "frac_issued": round(lrate_issued / lpeak, 4),
                        "frac_alone": round(lstm_alg / (lms_alone * 1e-3) / 1e12 / lpeak, 4) if lms_alone > 0 else None,
                        "frac_issued_alone": round(lstm_issued / (lms_alone * 1e-3) / 1e12 / lpeak, 4) if lms_alone > 0 else None,
                        "algorithmic_TFLOPs": round(lach, 2),
                        "frac_of_fp32_roof_algorithmic": round(lach / F32_MFMA_PEAK_TF, 4),
                        "frac_of_fp32_roof_algorithmic_alone": round(lstm_alg / (lms_alone * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4) if lms_alone > 0 else None,
                        "serial_steps_per_launch": T, "us_per_step": round(lms * 1e3 / T, 3), "us_per_step_alone": round(lms_alone * 1e3 / T, 3),
                        "handoff_floor_us": 0.25,
                        "frac_latency": round(0.25 / (lms_alone * 1e3 / T), 4) if lms_alone > 0 else None,
                        "tracks_per_serial_step": B,
                        "note": "3*T serially dependent steps per segment, every step a chain-wide hand-off; frac_latency = measured cross-CU "
                                "hand-off floor (tools/handoff_probe.hip) / step time",
                        "traffic": find_traffic(lname)})

        def stream_entry(key, name, nbytes, tneedle, per_lane_launches=False):
            # the streaming kernels of a track-batched context cover every lane in one launch (common.h LaneSet)
            nl = B if (per_lane_launches or not batched) else 1
            nbytes = nbytes * (B // nl)
            ms = stage_ms.get(key, 0.0) / nl
            ms_alone = stage_alone_ms.get(key, 0.0) / nl
            ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "bound": "hbm", "launches_per_step": nl, "launch_ms": round(ms, 4), "launch_ms_alone": round(ms_alone, 4),
                    "algorithmic_bytes_per_launch": nbytes, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4),
                    "frac_alone": round(nbytes / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_alone > 0 else None,
                    "traffic": find_traffic(tneedle)}
        wmode = os.environ.get("UMX_WIENER") or ("fused" if batched else "stats4")
        kernels.append(stream_entry("stft", "stft_kernel", byt["stft"], "stft_kernel"))
        if wmode == "fused":  # track-batched default: statistics, then gains + filter + inverse STFT frame in one kernel
            if not args.no_wiener:
                kernels.append(stream_entry("wiener", "wiener_stats4_kernel (+ finish4)", byt["wiener_stats4"], "wiener_stats4"))
            kernels.append(stream_entry("istft", "wiener_istft_kernel (filter + inverse STFT + overlap-add)", byt["wiener_istft"], "wiener_istft"))
        else:
            kernels += [stream_entry("wiener", "mixphase_kernel" if args.no_wiener else "wiener_{stats,finish,apply}_kernel (3 launches)",
                                     byt["mixphase"] if args.no_wiener else byt["wiener"], "wiener_apply", True),
                        stream_entry("istft", "istft_frames_kernel", byt["istft"], "istft_frames", True)]
            kernels.append(stream_entry("ola", "istft_ola_kernel", byt["ola"], "istft_ola"))
        # dominant = the largest share of a step by stand-alone time (the in-pipeline spans of the small per-track kernels
        # include whatever the other slot ran beside them)
        dominant = max(kernels, key=lambda kk: (kk["launch_ms_alone"] or kk["launch_ms"]) * kk["launches_per_step"])
        roofline = {kk: dominant.get(kk) for kk in ("kernel", "bound", "pipe", "achieved", "peak", "unit", "frac", "frac_issued", "traffic", "launch_ms",
                                                    "launch_ms_alone", "launches_per_step", "algorithmic_flops_per_launch",
                                                    "issued_flops_per_launch", "frac_alone", "frac_issued_alone", "algorithmic_TFLOPs",
                                                    "frac_of_fp32_roof_algorithmic", "frac_latency", "us_per_step", "us_per_step_alone") if kk in dominant}
        if roofline.get("bound") == "latency":
            roofline["bound_note"] = ("neither hbm nor mfma binds this kernel: 3*T serially dependent steps, each a chain-wide hand-off "
                                      "(SURVEY 8d); `peak` is the peak of the pipe it issues on, `frac` what the algorithmic flops make of it")
        roofline["traffic_source"] = os.path.relpath(traffic_src, ROOT) if traffic_src and traffic else None
        roofline["traffic_missing"] = traffic_missing or None  # kernels of this run with no row in that summary
        roofline["share_of_device_time"] = round(dominant["launch_ms"] * dominant["launches_per_step"] /
                                                 max(sum(kk["launch_ms"] * kk["launches_per_step"] for kk in kernels), 1e-9), 3)
        gemm_alone_ms = sum(kk["launch_ms_alone"] * kk["launches_per_step"] for kk in kernels[:4])
        gemm_alg = sum(kk["algorithmic_flops_per_launch"] * kk["launches_per_step"] for kk in kernels[:4])
        gemm_issued = sum(kk["issued_flops_per_launch"] * kk["launches_per_step"] for kk in kernels[:4])
        line = {
            "metric": "realtime-factor (audio-sec/wall-sec) UMX-L 4-stem, 60 s seg",
            "value": round(value, 2), "unit": "x realtime", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (activations as 2 x fp16 planes of the power-of-two-scaled row, integer weights exact in fp16, f32 accumulate on the fp16 matrix cores; fp32-grade: 1.4-2.2x the error of an fp32 evaluation, profiles/r06_v2_accuracy_vs_float64.txt)" if batched else "f32 (dense stack: operands as 3 bf16 terms on the bf16 matrix cores, f32 accumulate; recurrence on the fp32 vector pipe)",
            "data": "synthetic",
            "ms_per_step_min": round(min(serial), 3), "ms_per_step_median": round(float(np.median(serial)), 3),
            "ms_per_step_note": f"min / median over {len(serial)} further steps run one at a time (sync after each); ms_per_step is the mean of the timed region",
            "value_single_segment": single["value"] if single else None,
            "value_single_segment_pcie": single["value_pcie"] if single else None,
            "lone_segment_ms": single["lone_segment_ms"] if single else None,
            "lone_segment_ms_latency_context": single.get("lone_segment_ms_latency_context") if single else None,
            "checked_max_abs": single["checked_max_abs"] if single else None,
            "outputs_checked": (single["checked_max_abs"] is not None and single["checked_max_abs"] < 1e-5) if single else None,
            "value_pcie": (round(world * B * args.steps * seg_sec / dt_pcie, 2) if isinstance(dt_pcie, float) else None),
            "value_pcie_note": ("same steps with pinned host buffers: H2D audio + kernels + D2H stems inside the timed region, "
                                f"{B * 2 * N * 4 * 5 / 1e6:.0f} MB over PCIe per step" if isinstance(dt_pcie, float) else dt_pcie),
            "ms_per_step_pcie": round(dt_pcie / args.steps * 1e3, 3) if isinstance(dt_pcie, float) else None,
            "single_track": single,
            "config": {"workload": ("UMX-L single 60 s segment, " + ("vocals model only" if args.vocals_only else "4 stems") +
                                    (" (no Wiener)" if args.no_wiener else " + multichannel Wiener EM") +
                                    f" on 1 MI355X per rank, {B} independent track(s) per GPU (one 60 s segment of each per step); "
                                    "seeded synthetic 44.1 kHz stereo, synthetic UMX-L-shaped u8/u16 ggml weights"),
                       "hidden": H, "segment_samples": N, "frames": T, "stems": 4, "tracks_per_gpu": B,
                       "audio_seconds_per_step": B * seg_sec,
                       "lstm_kernel": ({"lstm_batch8_kernel": "batched, matrix cores, workgroups of 8 lanes x 64 hidden units" +
                                                              (", two octets per workgroup in turn" if B > 32 else "") + " (lstm_batch8_kernel)",
                                        "lstm_batch_kernel": "batched, matrix cores, groups of 16 lanes one launch after the other (lstm_batch_kernel)"}.get(lstm_kernel, lstm_kernel)
                                       if batched else "single-track, VALU (lstm_persistent_kernel)"),
                       "lstm": {0: "stepwise", 1: "persistent (sc1 hand-off)", 2: "persistent (intra-XCD hand-off)"}.get(lstm_mode, "?"),
                       "gemm": (flavour + (" (fp16 matrix cores, f32 accumulate: activations split once into 2 fp16 planes of the power-of-two "
                                           "scaled row, u8 weights exact in 1 plane (2 products), u16 weights exact in 2 planes, fp16(q) + remainder (3 products: a2 x remainder, 2^-22, is not formed), "
                                           "LDS-DMA staging, 256x256 tiles over all track lanes)" if flavour == "planes" else
                                           " (fp32 operands split into 3 bf16 terms while staged, 3 / 6 products)" if flavour == "bf16x3"
                                           else " MFMA")),
                       "weights_resident": ("expanded at load (f32 / bf16 planes)" if args.expanded_weights
                                            else "u8/u16 as in the file (BASELINE config 5)"),
                       "weight_bytes": weight_bytes,
                       "sharding": f"{world} x {B} independent tracks"},
            "roofline": roofline,
            "kernels": kernels,
            "stages_ms": {kk: round(v, 4) for kk, v in stage_ms.items()},
            "stages_ms_unpipelined": {kk: round(v, 4) for kk, v in stage_alone_ms.items()},
            "ms_per_step_unpipelined": round(min(serial), 3),
            "gemm_view": {"ms_alone_per_track": round(gemm_alone_ms / B, 3),
                          "algorithmic_TFLOPs": round(gemm_alg / (gemm_alone_ms * 1e-3) / 1e12, 1) if gemm_alone_ms > 0 else None,
                          "issued_bf16_TFLOPs": round(gemm_issued / (gemm_alone_ms * 1e-3) / 1e12, 1) if gemm_alone_ms > 0 else None,
                          "frac_of_bf16_mfma_peak": round(gemm_issued / (gemm_alone_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 3) if gemm_alone_ms > 0 else None},
            "outputs_finite": finite,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(pkg, wpath, args.cpu_sample_seconds, args.cpu_sample_seconds_1t)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        if args.track_seconds > 0:
            # what a umx-cli user sees (umx.cpp:99-295 on ONE GPU): a whole track through umx_hip_shift_inference, host buffers in and out.
            # carry = the reference's semantics (one lstm_data through the segments: the one-track engine, segments pipelined);
            # reset = UMX_FLAG_RESET_SEGMENTS, a DECLARED deviation (every segment from a zero state: the track's segments ride as
            # the lanes of one call of a 16-lane context) -- with the SDR of its stems against carry mode's next to the time.
            e1 = make_engine(1, False)
            Lt = int(args.track_seconds * 44100)
            twave = pkg.ggml.synth_audio(Lt, 99)
            ta = np.ascontiguousarray(twave.T).ravel()
            touts = [np.empty(2 * Lt, np.float32) for _ in range(4)]
            e1.separate_interleaved(ta, Lt, touts, shift_offset=4033)  # warm-up: allocates the track buffers
            best = min(e1.separate_interleaved(ta, Lt, touts, shift_offset=4033) for _ in range(3))
            nseg = -(-(Lt + 22050 - 4033) // int(0.75 * N))
            line["track"] = {"seconds_of_audio": args.track_seconds, "wall_ms": round(best * 1e3, 2),
                             "realtime_factor": round(args.track_seconds / best, 1), "segments": nseg, "mode": "carry (the reference's)",
                             "includes": "pageable H2D of the track, all segments pipelined, overlap-add on device, D2H of 4 stems"}
            e1.close()
            try:
                lanes = max(2, min(64, nseg))
                er = make_engine(lanes, False)
                routs = [np.empty(2 * Lt, np.float32) for _ in range(4)]
                er.separate_interleaved(ta, Lt, routs, flags=pkg.FLAG_RESET_SEGMENTS, shift_offset=4033)
                best_r = min(er.separate_interleaved(ta, Lt, routs, flags=pkg.FLAG_RESET_SEGMENTS, shift_offset=4033) for _ in range(3))
                sdr = [round(float(10 * np.log10(np.sum(touts[t].astype(np.float64) ** 2) /
                                                 max(np.sum((touts[t].astype(np.float64) - routs[t]) ** 2), 1e-300))), 2) for t in range(4)]
                line["track_reset_mode"] = {"wall_ms": round(best_r * 1e3, 2), "realtime_factor": round(args.track_seconds / best_r, 1),
                                            "lanes": lanes, "passes": -(-nseg // lanes),
                                            "sdr_db_vs_carry_per_stem": sdr,
                                            "note": "opt-in deviation from umx.cpp:167-171,226-227 (zero LSTM state per segment); synthetic weights and "
                                                    "audio: the SDR says how far the two modes are apart here, not what it is on music"}
                er.close()
            except Exception as e:  # noqa: BLE001
                line["track_reset_mode"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_track_mode(args, pkg, mg, dist, world, rank, local_rank, dev, wpath):
    """BASELINE config 4: ONE 600 s track, its 14 segments round-robin over the ranks with the exact per-layer (h, c)
    hand-off, weighted stems gathered on rank 0: the C++17 driver of include/umx_mgpu.h (RCCL send / recv on device
    pointers, no host bounce).  A step = the whole track, host buffers in and out."""
    import torch
    N = args.segment_samples
    eng = pkg.Engine.from_file(wpath, segment_samples=N, device=local_rank)
    secs = args.track_seconds or 600.0
    L = int(secs * 44100)
    wave = pkg.ggml.synth_audio(L, 99)
    ids = None
    if world > 1:  # RCCL rendezvous ids: made on rank 0, broadcast over the process group that is already up
        t = torch.zeros(pkg.MGPU_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            t = torch.frombuffer(bytearray(pkg.mgpu_unique_id()), dtype=torch.uint8).clone().to(dev)
        dist.broadcast(t, src=0)
        ids = bytes(t.cpu().numpy().tobytes())
    by_target = args.mode == "targets"
    drv = pkg.MultiGpuTrack(eng, rank, world, ids, by_target=by_target, loopback=args.loopback and world == 1)
    G = (4 if world % 4 == 0 else 2 if world % 2 == 0 else 1) if by_target else 1
    dd = dist if world > 1 else None
    ta = np.ascontiguousarray(wave.T).ravel()  # (2,L) interleaved, as umx_mgpu_separate_track takes it: no numpy work in the timed step
    touts = [np.empty(2 * L, np.float32) for _ in range(4)] if rank == 0 else None
    res = [touts]

    def step():
        drv.separate_interleaved(ta, L, touts, shift_offset=4033)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
    steps, warmup = max(1, min(args.steps, 4)), max(1, min(args.warmup, 1))
    dt = mg.timed_region(step, fence, steps, warmup, dist=dd, world=world, device=dev)
    if rank == 0:
        nseg = -(-(L + 22050 - 4033) // int(0.75 * N))
        line = {"metric": "realtime-factor (audio-sec/wall-sec) UMX-L 4-stem, 60 s seg", "value": round(steps * secs / dt, 2),
                "unit": "x realtime", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32 (dense stack: operands as 3 bf16 terms on the bf16 matrix cores, f32 accumulate; recurrence on the fp32 vector pipe)", "data": "synthetic",
                "config": {"workload": f"UMX-L full-track segmented inference: one {secs:g} s track, {nseg} segments over {world} MI355X as "
                                       f"{G} target group(s) x {world // G} segment-pipeline stage(s), exact LSTM state carry (per-layer (h, c) by "
                                       "RCCL send/recv between the engines' HBM state buffers, sends on their own stream and communicator), "
                                       + ("target magnitudes point to point to the rank that runs the Wiener filter of the segment, " if G > 1 else "")
                                       + "weighted stems gathered and overlap-added on rank 0; host buffers in and out (BASELINE config 4)",
                           "hidden": args.hidden, "segment_samples": N, "segments": nseg,
                           "parallelism": (f"{G} target groups x {world // G} stages" if by_target else f"segments over {world} ranks, carry mode") +
                                          "; per target at most 3 segments are in flight (one per LSTM layer)",
                           "rccl": drv.stats()},
                "outputs_finite": bool(all(np.isfinite(r).all() for r in res[0]))}
        print(json.dumps(line), flush=True)
    drv.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
