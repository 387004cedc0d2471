/*
 * umx_mgpu.h -- C-ABI of the multi-GPU track driver (C++17 host, libumx_mgpu.so): BASELINE config 4,
 * "UMX-L full-track segmented inference, 4 stems x segments sharded over 2/4/8 MI355X via RCCL, overlap-add gather".
 *
 * Replaces split_inference / shift_inference (umx.cpp:99-295) for ONE track over `world` GPUs of one node, one
 * process (rank) per GPU, EXACTLY.  Two axes (host/shard_plan.h), world = G target groups x P pipeline stages:
 *   * segments (carry mode, G = 1): the reference carries every LSTM chain's (h, c) across segments (umx.cpp:167-171,
 *     lstm.cpp:139-161: SURVEY F3), and layer l of segment s needs only layer l's state of segment s-1.  Segment s runs
 *     on stage s % P through the phased engine API (include/umx_hip.h: umx_hip_segment_begin_device / _lstm_layer /
 *     ...); a layer's state goes stage to stage with ncclSend / ncclRecv DIRECTLY between the engines' HBM state
 *     buffers (no host bounce, no host synchronisation).
 *   * source models (target mode, G = gcd(world, 4)): the four target networks of umx_inference (inference.cpp:70-186)
 *     are independent until wiener_filter (inference.cpp:192-193).  Rank (g, p) runs the targets t % G == g of the
 *     segments s % P == p; no LSTM state ever crosses between groups; the target magnitudes (2 x T x 2049 floats per
 *     target) go point to point to the one rank of the stage that filters this segment (rotating with the segment
 *     index), which runs the Wiener EM + inverse STFT.  8 GPUs = 4 targets x a 2-stage segment pipeline.
 * Every rank that owns stems weights them on the device (umx.cpp:246) and sends them to rank 0, where they are added in
 * segment order and normalised (umx.cpp:234-273).  All traffic is point to point over xGMI; there is no data-path
 * collective (one 4-byte all-reduce per track carries the ranks' status words).  The schedule is the one of
 * umx_split_inference_carry / _targets (include/umx_host.h), which the CPU tests drive over gloo; the result equals
 * umx_hip_split_inference / _shift_inference on one GPU bit for bit.
 *
 * Streams and communicators (why the schedule cannot deadlock whatever RCCL's send semantics): receives of LSTM state
 * sit on the engine's own stream in front of the layer that needs them; SENDS sit on a stream of their own behind an
 * event of the layer that produced them, and use a different communicator than the receives of the same rank (ring
 * edges are coloured, shard_plan.h) -- a send therefore never stands in front of anything another rank is waiting for,
 * and the cross-rank dependencies are exactly the data dependencies (segment s-1 before s), which are acyclic.  Target
 * magnitudes and stems travel on two further streams / communicators, their operations issued in segment order by
 * every rank.
 *
 * Failure: a persistent-LSTM timeout on any rank (another process holding the CUs) is found by every rank through the
 * status all-reduce before rank 0 hands anything out, and the whole track is run again once, the rank concerned on its
 * per-step LSTM driver (bit-identical).  Any other error aborts this rank's communicators (ncclCommAbort) and returns.
 * No rank waits for ever on a peer: every host-side wait of a track polls its streams against a deadline
 * (environment UMX_MGPU_TIMEOUT_MS, default 120000) and asks the communicators for asynchronous errors
 * (ncclCommGetAsyncError); on expiry or on an error the rank aborts its communicators -- its queued transfers end, its
 * streams drain -- and umx_mgpu_separate_track returns UMX_ERR_TIMEOUT ("watchdog: ...") or UMX_ERR_HIP.  A driver whose
 * communicators were aborted refuses further tracks; destroy it and create a new one (a new rendezvous).
 */
#ifndef UMX_MGPU_H
#define UMX_MGPU_H

#include "umx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UMX_MGPU_ID_BYTES 640 /* five ncclUniqueId: state ring (three edge colours), stem gather, target magnitudes */

#define UMX_MGPU_BY_TARGET 0x1u /* shard by source model as well: G = gcd(world, 4) target groups x world / G stages */
#define UMX_MGPU_LOOPBACK 0x2u  /* world == 1 only: a one-rank communicator, and every state hop, magnitude exchange and
                                   stem gather goes through a grouped RCCL self send + receive on the same streams as in a
                                   real run (the outgoing buffer is poisoned in between).  Lets a one-GPU box execute the
                                   RCCL path; environment UMX_MGPU_LOOPBACK=1 sets it too. */

typedef struct umx_mgpu umx_mgpu;

/* Rank 0 creates the rendezvous ids and hands the bytes to every rank out of band (bench.py: torch.distributed
 * broadcast; a launcher: a file or an environment variable). */
int umx_mgpu_unique_id(char id[UMX_MGPU_ID_BYTES], char *err);
/* ctx: this rank's engine -- a single-track context (its LSTM workgroups share compute units with RCCL's kernels; the
 * track-batched kernels need every CU to themselves).  world == 1 needs no ids (pass NULL) and, without
 * UMX_MGPU_LOOPBACK, never touches RCCL.  Reserves a few CUs for RCCL at the engine's admission gate
 * (umx_hip_gate_reserve; environment UMX_MGPU_RESERVE_CUS, default 16). */
int umx_mgpu_create(umx_mgpu **out, umx_hip_ctx *ctx, int rank, int world, const char id[UMX_MGPU_ID_BYTES], char *err);
int umx_mgpu_create_ex(umx_mgpu **out, umx_hip_ctx *ctx, int rank, int world, const char id[UMX_MGPU_ID_BYTES], unsigned mgpu_flags,
                       char *err);
void umx_mgpu_destroy(umx_mgpu *m);
/* The whole track: audio_host (2,length) on every rank (each rank uploads only the segments it runs), out_host[4]
 * (2,length) written on rank 0.  shift_offset < 0: split_inference; >= 0: shift_inference with that offset
 * (umx.cpp:115; the reference's unseeded rand() gives 4033).  Collective over the ranks: every rank must call it.
 * Sharded by segment only, or also by target if the driver was created with UMX_MGPU_BY_TARGET. */
int umx_mgpu_separate_track(umx_mgpu *m, const float *audio_host, int length, int shift_offset, float *const out_host[4],
                            unsigned flags, char *err);
/* {RCCL operations queued by the last track, of them state hops, magnitude transfers, stem transfers, tracks retried} */
int umx_mgpu_stats(const umx_mgpu *m, long long out5[5]);

#ifdef __cplusplus
}
#endif
#endif /* UMX_MGPU_H */
