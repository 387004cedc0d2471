/*
 * umx_mgpu.h -- C-ABI of the multi-GPU track driver (C++17 host, libumx_mgpu.so): BASELINE config 4,
 * "UMX-L full-track segmented inference, 4 stems x segments sharded over 2/4/8 MI355X via RCCL, overlap-add gather".
 *
 * Replaces split_inference / shift_inference (umx.cpp:99-295) for ONE track over `world` GPUs of one node, one
 * process (rank) per GPU, EXACTLY: the reference carries every LSTM chain's (h, c) across segments (umx.cpp:167-171,
 * lstm.cpp:139-161: SURVEY F3), and layer l of segment s needs only layer l's state of segment s-1.  Segment s runs
 * on rank s % world through the phased engine API (include/umx_hip.h: umx_hip_segment_begin_device / _lstm_layer /
 * _end_device); the 4 x 4 x hidden/2 floats of a layer's state go rank to rank with ncclSend / ncclRecv DIRECTLY
 * between the engines' HBM state buffers, on the engine's own stream (no host bounce, no host synchronisation);
 * every rank weights its stems on the device (umx.cpp:246) and sends them to rank 0 over a second communicator and
 * stream, where they are added in segment order and normalised (umx.cpp:234-273).  All traffic is point to point over
 * xGMI; there is no collective.  The schedule is the one of umx_split_inference_carry (include/umx_host.h), which the
 * CPU tests drive over gloo; the result equals umx_hip_split_inference / _shift_inference on one GPU bit for bit.
 */
#ifndef UMX_MGPU_H
#define UMX_MGPU_H

#include "umx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UMX_MGPU_ID_BYTES 256 /* two ncclUniqueId (state ring, stem gather) */

typedef struct umx_mgpu umx_mgpu;

/* Rank 0 creates the rendezvous ids and hands the bytes to every rank out of band (bench.py: torch.distributed
 * broadcast; a launcher: a file or an environment variable). */
int umx_mgpu_unique_id(char id[UMX_MGPU_ID_BYTES], char *err);
/* ctx: this rank's engine (one track lane is used).  world == 1 needs no ids (pass NULL) and never touches RCCL. */
int umx_mgpu_create(umx_mgpu **out, umx_hip_ctx *ctx, int rank, int world, const char id[UMX_MGPU_ID_BYTES], char *err);
void umx_mgpu_destroy(umx_mgpu *m);
/* The whole track: audio_host (2,length) on every rank (each rank uploads only the segments it runs), out_host[4]
 * (2,length) written on rank 0.  shift_offset < 0: split_inference; >= 0: shift_inference with that offset
 * (umx.cpp:115; the reference's unseeded rand() gives 4033).  Collective over the ranks: every rank must call it. */
int umx_mgpu_separate_track(umx_mgpu *m, const float *audio_host, int length, int shift_offset, float *const out_host[4],
                            unsigned flags, char *err);

#ifdef __cplusplus
}
#endif
#endif /* UMX_MGPU_H */
