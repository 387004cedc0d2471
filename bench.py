#!/usr/bin/env python
"""bench.py -- realtime factor of UMX-L 4-stem separation on 60 s segments (BASELINE.json metric).

A "step" = one pass of the hot path (umx_inference, inference.cpp:12-207) over one synthetic 60 s
stereo segment: STFT -> 4 x [fc1/bn/tanh -> 3-layer BiLSTM -> fc2 -> fc3 -> mask] -> Wiener EM ->
4 x iSTFT, with the input already resident in HBM and the 4 stems left in HBM.  Consecutive steps
are consecutive segments of one track: the streaming LSTM state carries over (umx.cpp:167-171).
The default workload is BASELINE config 3 (4 stems + Wiener, the full umx_inference); --no-wiener
gives config 2.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank separates its own
independent segments (weak scaling; the path shards by segment/track, no data-path collective);
barrier + synchronize on both sides, MAX over ranks, rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

SEG = 60 * 44100
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def algorithmic_work(T, H, wiener=True):
    """Per-segment algorithmic flops / bytes per stage."""
    NB, KX, NOUT = 2049, 2974, 4098
    gemm = {
        "fc1": 2.0 * T * KX * H * 4,
        "lstm_ih": 2.0 * T * H * (4 * H) * 4,          # per layer, both directions, 4 targets
        "fc2": 2.0 * T * (2 * H) * H * 4,
        "fc3_mask": 2.0 * T * H * NOUT * 4,
    }
    rec = 2.0 * T * (H // 2) * (2 * H) * 2 * 4         # per layer: W_hh.h, 2 dirs, 4 targets
    bytes_ = {
        "stft": 4.0 * (2 * T * 1024 + 2 * T * NB * 2 + 2 * T * NB + T * 2976),
        "wiener": 4.0 * (4 * (2 * T * NB * 3) + (2 * T * NB * 2 + 4 * 2 * T * NB) + 4 * 2 * T * NB * 2) if wiener
        else 4.0 * (2 * T * NB * 2 + 4 * 2 * T * NB + 4 * 2 * T * NB * 2),
        "istft": 4.0 * (4 * 2 * T * NB * 2 + 4 * T * 4096 * 2),
        "ola": 4.0 * (4 * T * 4096 * 2 + 4 * 2 * T * 1024),
    }
    return gemm, rec, bytes_


def cpu_baseline(pkg, hidden, weights_path, seconds_audio=6.0, threads=None):
    """Reference-equivalent CPU path (oracle, reference flag set -O3 -march=native -ffast-math) on a
    bounded sample of the same workload: the first `seconds_audio` of the synthetic track as one
    segment (same per-frame cost as a 60 s segment; LSTM state zero).  NOT the Eigen binary."""
    po = ge.load_oracle()
    threads = threads or os.cpu_count()
    po.set_num_threads(threads, fast=True)
    om = po.Model.load(weights_path, fast=True)
    n = int(seconds_audio * 44100)
    wave = pkg.ggml.synth_audio(n, seed=0)
    t0 = time.time()
    po.umx_inference(om, wave)
    dt = time.time() - t0
    return {"value": round(seconds_audio / dt, 4), "unit": "x realtime (audio-sec / wall-sec)", "cores": threads,
            "kind": "port",
            "sample": f"first {seconds_audio:g} s of the synthetic track as one segment ({n // 1024 + 1} frames), "
                      f"4 stems + Wiener, {dt:.1f} s wall; Eigen-equivalent restatement (oracle/, -O3 -march=native "
                      f"-ffast-math -fopenmp, per-timestep GEMV LSTM), not the Eigen binary"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--segment-samples", type=int, default=SEG)
    ap.add_argument("--no-wiener", action="store_true")
    ap.add_argument("--stepwise-lstm", action="store_true")
    ap.add_argument("--safe-lstm", action="store_true", help="persistent LSTM kernel without the intra-XCD hand-off")
    ap.add_argument("--lstm-profile", action="store_true", help="print per-phase shader-clock counters of the LSTM kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-seconds", type=float, default=15.0)
    ap.add_argument("--serial", action="store_true", help="sync after every segment (no cross-segment pipelining)")
    ap.add_argument("--tracks", type=int, default=1,
                    help="independent tracks per GPU run together (track lanes, 1..16): one step = one 60 s segment of EVERY "
                         "track; > 1 selects the batched matrix-core LSTM kernel (SURVEY 8f-4)")
    ap.add_argument("--batched-lstm", action="store_true", help="use the batched LSTM kernel also with --tracks 1")
    ap.add_argument("--track-seconds", type=float, default=0.0,
                    help="also time a whole track of this length through umx_hip_shift_inference (host buffers in and "
                         "out, PCIe included; BASELINE config 4 on one GPU) and report it as 'track'")
    ap.add_argument("--gemm", choices=["bf16x3", "f32"], default=None,
                    help="dense-stack GEMM flavour: bf16x3 (default; three-term bf16 split, fp32-class accuracy) or f32 MFMA")
    ap.add_argument("--expanded-weights", action="store_true",
                    help="expand the u8/u16 weights at load time (fp32 / three bf16 planes in HBM) instead of keeping "
                         "them quantised in HBM with dequantisation inside the kernels (the default, BASELINE config 5)")
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
    H, N = args.hidden, args.segment_samples
    # synthetic UMX-L-shaped weights in the reference's file format (u8/u16 + scale/offset); every
    # rank writes its own copy (seeded, identical) to avoid a file-system race
    tmpdir = tempfile.mkdtemp(prefix=f"umx_bench_r{rank}_")
    wpath = os.path.join(tmpdir, "ggml-model-synth-u8.bin")
    pkg.ggml.write_model(wpath, pkg.ggml.synth_weights(H, seed=0), H, compress=False)
    B = args.tracks
    eng = pkg.Engine.from_file(wpath, segment_samples=N, device=local_rank, quantised_resident=not args.expanded_weights,
                               gemm=args.gemm, tracks=B, lstm_batched=args.batched_lstm)
    T = eng.T

    # each rank (and each track lane): its own track
    audios = [torch.from_numpy(np.ascontiguousarray(pkg.ggml.synth_audio(N, seed=rank * 16 + b).T).ravel()).to(dev)
              for b in range(B)]  # (2,n) interleaved, in HBM
    audio = audios[0]
    # two output sets: consecutive segments are in flight together (two pipeline slots) and must not share stems
    out_sets = [[torch.empty(2 * N, dtype=torch.float32, device=dev) for _ in range(4 * B)] for _ in range(2)]
    flags = (pkg.FLAG_NO_WIENER if args.no_wiener else 0) | (pkg.FLAG_LSTM_STEPWISE if args.stepwise_lstm else 0) | \
        (pkg.FLAG_LSTM_FORCE_SAFE if args.safe_lstm else 0) | (pkg.FLAG_LSTM_PROFILE if args.lstm_profile else 0)
    ptr_sets = [[o.data_ptr() for o in st_] for st_ in out_sets]
    nstep = [0]

    aptrs = [a.data_ptr() for a in audios]

    def step():
        if B == 1:
            eng.infer_segment_device(audio.data_ptr(), N, ptr_sets[nstep[0] & 1], flags)
        else:
            eng.infer_batch_ptrs(aptrs, [N] * B, ptr_sets[nstep[0] & 1], flags)
        nstep[0] += 1
        if args.serial:
            eng.sync()

    def fence():
        eng.sync()
        torch.cuda.synchronize()

    import importlib
    mg = importlib.import_module("umx_cpp_amd.multigpu")
    # W untimed warm-up steps, barrier + synchronize on both sides of exactly K steps, MAX over ranks
    dt = mg.timed_region(step, fence, args.steps, args.warmup, dist=dist if world > 1 else None, world=world, device=dev)
    # per-stage device time from hipEvents on the engine's own streams.  (a) the last two segments of the
    # timed region (one per pipeline slot; consecutive segments overlap there, so a span includes the other
    # slot's kernels -- this is the duration rocprofv3 reports for the same command); (b) three extra
    # segments run one at a time after the timed region (the kernel alone on the chip).
    prof_pipelined = eng.lstm_profile() if args.lstm_profile else None
    st = [eng.stage_times(slot=i) for i in (0, 1)]
    st = [d for d in st if d]
    stage_ms = {k: sum(d[k] for d in st) / len(st) for k in st[0]} if st else {}
    serial = []
    for _ in range(3):
        t1 = time.perf_counter()
        step()
        eng.sync()
        serial.append((time.perf_counter() - t1) * 1e3)
        stage_alone_ms = eng.stage_times()
    serial_ms = min(serial)
    finite = bool(all(torch.isfinite(o).all().item() for st_ in out_sets for o in st_))

    if rank == 0:
        seg_sec = N / 44100.0
        value = world * B * args.steps * seg_sec / dt
        gemm, rec, byt = algorithmic_work(T, H, not args.no_wiener)
        gemm = {k: v * B for k, v in gemm.items()}  # a stage's time spans every track lane
        rec *= B
        byt = {k: v * B for k, v in byt.items()}
        lstm_ms = sum(stage_ms.get(f"lstm_rec{l}", 0.0) for l in range(3))
        gemm_ms = stage_ms.get("fc1", 0) + stage_ms.get("fc2", 0) + stage_ms.get("fc3_mask", 0) + \
            sum(stage_ms.get(f"lstm_ih{l}", 0.0) for l in range(3))
        gemm_flops = gemm["fc1"] + 3 * gemm["lstm_ih"] + gemm["fc2"] + gemm["fc3_mask"]
        # dominant kernel by device time
        lstm_alone_ms = sum(stage_alone_ms.get(f"lstm_rec{l}", 0.0) for l in range(3))
        if lstm_ms >= gemm_ms:
            # The recurrence is neither HBM- nor MFMA-bound: it is 3*T serially dependent steps whose floor is
            # the cross-CU hand-off latency.  The schema wants hbm|mfma, so its fp32 FMA
            # work is priced against the f32 peak; the HBM and latency views are given next to it.
            per_launch_flops = rec
            per_launch_ms = lstm_ms / 3
            ach = per_launch_flops / (per_launch_ms * 1e-3) / 1e12
            whh_el = 4.0 if args.expanded_weights else 1.0  # W_hh resident as fp32 or as the file's u8
            alg_bytes = 4.0 * (4 * T * 4 * H + 8 * 2 * H + 4 * T * H) + whh_el * 8 * (H // 2) * (2 * H)  # P, b_hh, out, W_hh
            traffic = None
            if H == 1024 and T == 2584 and eng.lstm_mode() >= 1:
                # profiles/r01_v5_pmc_fetch_write_per_kernel.csv (u8 W_hh) / r01_v4 (fp32 W_hh): rocprofv3 --pmc
                # FETCH_SIZE / WRITE_SIZE (separate passes), KB per launch; FETCH_SIZE x2 on gfx950
                # (MI355X_MICROARCH.md HBM section)
                traffic = (2 * (99512.3 if args.expanded_weights else 86940.7) + 41456.0) * 1024
            roofline = {"kernel": "lstm_persistent_kernel<64>" if eng.lstm_was_persistent() else "lstm_step_kernel",
                        "bound": "mfma", "achieved": round(ach, 3), "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(ach / F32_MFMA_PEAK_TF, 5), "traffic": traffic,
                        "launch_ms": round(per_launch_ms, 4), "launch_ms_alone": round(lstm_alone_ms / 3, 4),
                        "algorithmic_flops_per_launch": per_launch_flops, "algorithmic_bytes_per_launch": alg_bytes,
                        "hbm_view": {"achieved_GBs": round(alg_bytes / (per_launch_ms * 1e-3) / 1e9, 1),
                                     "peak_GBs": HBM_PEAK_GBS,
                                     "frac": round(alg_bytes / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                        "latency_view": {"serial_steps_per_launch": T, "us_per_step": round(per_launch_ms * 1e3 / T, 3),
                                         "us_per_step_alone": round(lstm_alone_ms * 1e3 / (3 * T), 3),
                                         "handoff_floor_us": 0.25}}
        else:
            ach = gemm_flops / (gemm_ms * 1e-3) / 1e12
            roofline = {"kernel": "gemm_tn_kernel", "bound": "mfma", "achieved": round(ach, 3),
                        "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TF, 5),
                        "traffic": None}
        gemm_tf = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
        # the GEMM kernels against their own roof, stand-alone durations: fp32-equivalent (algorithmic) flops, and for
        # the bf16x3 flavour the six bf16 products it actually issues per fp32 product, against the dense bf16 peak
        gemm_alone_ms = stage_alone_ms.get("fc1", 0) + stage_alone_ms.get("fc2", 0) + stage_alone_ms.get("fc3_mask", 0) + \
            sum(stage_alone_ms.get(f"lstm_ih{l}", 0.0) for l in range(3))
        flavour = args.gemm or os.environ.get("UMX_GEMM", "bf16x3")
        gemm_view = None
        if gemm_alone_ms > 0:
            alg_tf = gemm_flops / (gemm_alone_ms * 1e-3) / 1e12
            gemm_view = {"kernel": "gemm_bf16x3_kernel" if flavour == "bf16x3" else "gemm_tn_kernel", "bound": "mfma",
                         "ms_alone": round(gemm_alone_ms, 3), "algorithmic_TFLOPs": round(alg_tf, 1),
                         "frac_of_f32_mfma_peak": round(alg_tf / F32_MFMA_PEAK_TF, 3)}
            if flavour == "bf16x3":
                gemm_view.update({"issued_bf16_TFLOPs": round(6 * alg_tf, 1), "bf16_mfma_peak": 2500.0,
                                  "frac_of_bf16_mfma_peak": round(6 * alg_tf / 2500.0, 3)})
        stream_ms = sum(stage_ms.get(k, 0.0) for k in ("stft", "wiener", "istft", "ola"))
        stream_gbs = sum(byt.values()) / (stream_ms * 1e-3) / 1e9 if stream_ms > 0 else None
        line = {
            "metric": "realtime-factor (audio-sec/wall-sec) UMX-L 4-stem, 60 s seg",
            "value": round(value, 2), "unit": "x realtime", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("UMX-L single 60 s segment, 4 stems" +
                                    (" (no Wiener)" if args.no_wiener else " + multichannel Wiener EM") +
                                    " on 1 MI355X per rank; seeded synthetic 44.1 kHz stereo, synthetic "
                                    "UMX-L-shaped u8/u16 ggml weights"),
                       "hidden": H, "segment_samples": N, "frames": T, "stems": 4,
                       "lstm": {0: "stepwise", 1: "persistent (sc1 hand-off)", 2: "persistent (intra-XCD hand-off)"}.get(
                           eng.lstm_mode(), "?"),
                       "gemm": ((args.gemm or os.environ.get("UMX_GEMM", "bf16x3")) +
                                (" (fp32 operands split into 3 bf16 terms, 6 products, f32 accumulate: error below fp32 rounding)"
                                 if (args.gemm or os.environ.get("UMX_GEMM", "bf16x3")) == "bf16x3" else " MFMA")),
                       "weights_resident": ("expanded at load (f32 / bf16 planes)" if args.expanded_weights
                                            else "u8/u16 as in the file (dequantised in the kernels)"),
                       "weight_bytes": eng.weight_bytes(),
                       "tracks_per_gpu": B, "lstm_kernel": "batched (matrix cores)" if eng.lstm_is_batched() else "single-track (VALU)",
                       "sharding": f"{world} x {B} independent tracks (one 60 s segment of each per step)"},
            "roofline": roofline,
            "stages_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stages_ms_unpipelined": {k: round(v, 4) for k, v in stage_alone_ms.items()},
            "ms_per_segment_unpipelined": round(serial_ms, 3),
            "gemm_tflops": round(gemm_tf, 2) if gemm_tf else None,
            "gemm_view": gemm_view,
            "streaming_gbs": round(stream_gbs, 1) if stream_gbs else None,
            "outputs_finite": finite,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(pkg, H, wpath, args.cpu_sample_seconds)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        if args.track_seconds > 0:
            Lt = int(args.track_seconds * 44100)
            twave = pkg.ggml.synth_audio(Lt, 99)
            ta = np.ascontiguousarray(twave.T).ravel()
            touts = [np.empty(2 * Lt, np.float32) for _ in range(4)]
            eng.separate_interleaved(ta, Lt, touts, shift_offset=4033)  # warm-up: allocates the track buffers
            best = min(eng.separate_interleaved(ta, Lt, touts, shift_offset=4033) for _ in range(3))
            line["track"] = {"seconds_of_audio": args.track_seconds, "wall_ms": round(best * 1e3, 2),
                             "realtime_factor": round(args.track_seconds / best, 1),
                             "segments": -(-(Lt + 22050 - 4033) // int(0.75 * N)),
                             "includes": "pageable H2D of the track, all segments pipelined, overlap-add on device, D2H of 4 stems"}
        print(json.dumps(line), flush=True)
        if args.lstm_profile:
            pr = eng.lstm_profile()
            if prof_pipelined is not None and not args.serial:
                pp = prof_pipelined.reshape(-1)[:48].reshape(3, 2, 8)
                for layer in range(3):
                    for w in range(2):
                        c = pp[layer, w]
                        n = max(int(c[4]), 1)
                        print(f"# pipelined lstm layer {layer} wave {w}: cycles/step poll {c[0] / n:.0f} dot {c[1] / n:.0f} "
                              f"barrier {c[2] / n:.0f} gates {c[3] / n:.0f} failed-polls/step {c[5] / n:.2f} (steps {int(c[4])})", file=sys.stderr)
            for layer in range(3):
                for w in range(2):
                    c = pr[layer, w]
                    n = max(int(c[4]), 1)
                    print(f"# lstm layer {layer} wave {w}: cycles/step poll {c[0] / n:.0f} dot {c[1] / n:.0f} "
                          f"barrier {c[2] / n:.0f} gates {c[3] / n:.0f} failed-polls/step {c[5] / n:.2f} (steps {int(c[4])})"
                          f" [poll: sleep {c[6] / n:.0f} first-loads {c[7] / n:.0f}]", file=sys.stderr)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
