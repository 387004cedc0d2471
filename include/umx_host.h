/*
 * umx_host.h -- C-ABI of the C++17 host side that surrounds the segment engine: the ggml weight
 * loader, WAV in/out and the segmented-apply drivers.  Pure host code (no HIP): it is what the
 * reference keeps in src/model.cpp, src/dsp.cpp (load_audio / write_audio_file) and umx.cpp, and
 * it talks to the device only through include/umx_hip.h.
 *
 * Every function returns an int status (0 = ok) and never exits the process; `err`, when given,
 * points to a buffer of at least UMX_ERRLEN bytes that receives a message.
 */
#ifndef UMX_HOST_H
#define UMX_HOST_H

#include "umx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UMX_ERRLEN 256
#define UMX_HOST_ERR_IO 10      /* cannot open / read / write a file */
#define UMX_HOST_ERR_FORMAT 11  /* bad magic, truncated record, unknown tensor name, wrong shape */
#define UMX_HOST_ERR_AUDIO 12   /* unsupported sample rate / channel count / encoding */
#define UMX_HOST_ERR_BACKEND 13 /* the segment backend returned an error */

#define UMX_SAMPLE_RATE 44100          /* dsp.hpp:16 SUPPORTED_SAMPLE_RATE */
#define UMX_SEGMENT_SAMPLES 2646000    /* inference.hpp:13 SEGMENT_LEN_SECS * 44100 (umx.cpp:156-157) */
#define UMX_MAX_SHIFT_SAMPLES 22050    /* inference.hpp:14 MAX_SHIFT_SECS * 44100 (umx.cpp:112-113) */

/* ---- weight file: replaces load_umx_model (src/model.cpp:42-574, declared src/model.hpp:58).
 * Reads a gzip'ed (or plain) ggml-style file entirely in memory (the reference inflates to
 * ./temp.decompressed in 128-byte reads, model.cpp:56-84) and keeps every tensor as stored
 * (u8/u16 + scale/offset); dequantisation happens where the tensor is consumed. */
typedef struct umx_model umx_model;
int umx_model_load(const char *path, umx_model **out, char *err);
void umx_model_free(umx_model *m);
int umx_model_hidden(const umx_model *m);                      /* model.cpp:109-114 */
int umx_model_n_tensors(const umx_model *m);                   /* 172 for a complete file (README.md:191) */
const umx_tensor_view *umx_model_views(const umx_model *m);    /* n_tensors entries, valid while m lives */
size_t umx_model_data_bytes(const umx_model *m);               /* the "131.93 MB" of model.cpp:566-568 */
float umx_model_load_progress(const umx_model *m);             /* model.hpp:54, 1.0 when loaded */
/* fp32 copy of one tensor, dequantised exactly like model.cpp:610-616 / 656-662; returns numel or <0 */
long umx_model_dequantize(const umx_model *m, int target, const char *name, float *dst, size_t capacity);

/* ---- audio files: replaces load_audio / write_audio_file (src/dsp.cpp:18-101, libnyquist).
 * load: RIFF/WAVE PCM 16/24/32-bit int or 32-bit float (plain or WAVE_FORMAT_EXTENSIBLE), mono or
 * stereo, 44100 Hz only (dsp.cpp:27-33 exits on another rate; here UMX_HOST_ERR_AUDIO).  Returns a
 * malloc'ed (2,n) interleaved stereo buffer, mono duplicated into both channels (dsp.cpp:52-60). */
int umx_wav_load(const char *path, float **audio_out, int *n_frames_out, int *channels_in_file, char *err);
void umx_wav_free(float *audio);
/* write: stereo 32-bit IEEE float WAV, 44100 Hz (dsp.cpp:97-99 {channels, PCM_FLT, ...}) */
int umx_wav_write_f32(const char *path, const float *audio, int n_frames, char *err);

/* ---- segmented apply: replaces split_inference / shift_inference (umx.cpp:99-295).
 * The backend is the per-segment call (umx_inference, umx.cpp:226-227): audio (2,n) -> out[4] (2,n);
 * reset is called once per track where the reference creates its 4 zeroed lstm_data
 * (umx.cpp:167-171).  host/umx_cli.cpp adapts a umx_hip_ctx to this; tests plug the oracle. */
typedef int (*umx_segment_fn)(void *user, const float *audio, int n, float *const out[4]);
typedef int (*umx_reset_fn)(void *user);
typedef struct umx_backend
{
    umx_segment_fn segment;
    umx_reset_fn reset; /* may be NULL */
    void *user;
} umx_backend;

/* 60 s segments, stride = int(0.75 * segment) (umx.cpp:181), triangular weights (umx.cpp:197-206),
 * weighted overlap-add and division by the weight sum (umx.cpp:234-273).  Deliberate deviation:
 * sum_weight is fully zero-initialised (the reference writes it out of bounds / leaves it
 * uninitialised for tracks < 29.6 s, umx.cpp:197-204 -- SURVEY F4).  progress (optional) receives
 * the reference's inference_progress value after each segment (umx.cpp:229). */
int umx_split_inference(const umx_backend *be, const float *audio, int length, int segment_samples,
                        float *const out[4], void (*progress)(float, void *), void *progress_user, char *err);
/* umx.cpp:99-150: delay by `offset` inside a zero buffer of length+22050-offset samples, split, trim.
 * offset < 0 -> the reference's rand() % 22050 (unseeded glibc rand(): UMX_REFERENCE_SHIFT = 4033). */
int umx_shift_inference(const umx_backend *be, const float *audio, int length, int segment_samples,
                        int offset, float *const out[4], void (*progress)(float, void *), void *progress_user,
                        char *err);

/* ---- one track over several ranks, EXACT ("carry" mode).  split_inference's segments (umx.cpp:214-227) go
 * round-robin over `world` ranks; the reference carries every LSTM chain's (h, c) from one segment into the next
 * (umx.cpp:167-171, lstm.cpp:139-161: SURVEY F3) and layer l of segment s needs nothing else of segment s-1, so rank
 * s % world runs segment s phase by phase: before LSTM layer l it receives that layer's state from the rank that ran
 * segment s-1 and afterwards sends its own on.  Rank 0 gathers the weighted stems in segment order (each output
 * sample has at most two contributors), adds them exactly like umx.cpp:234-260 and normalises.  All traffic is point
 * to point; there is no collective.  Bit-identical to umx_split_inference on one rank.
 * The backend is the phased form of the per-segment call (include/umx_hip.h: umx_hip_segment_begin / _lstm_layer /
 * _end + umx_hip_stream_{get,set}_layer); the transport moves float buffers between ranks and must NOT block a send until
 * the matching receive is posted (buffered or non-blocking sends): with two ranks, rank 0 sends layer 1 of segment 0
 * before it receives layer 0 of segment 2, while rank 1 sends layer 0 of segment 1 before it receives layer 1 of segment
 * 0 -- rendezvous sends would deadlock.  This host-buffer form is what the CPU tests drive
 * (torch.distributed gloo through callbacks); host/mgpu.cpp is the same schedule with device buffers over RCCL. */
typedef struct umx_phased_backend
{
    int (*begin)(void *user, const float *audio, int n);     /* front of a segment: STFT, fc1, W_ih of layer 0 */
    int (*layer)(void *user, int l);                          /* LSTM layer l (0, 1, 2 in order) */
    int (*end)(void *user, float *const out[4]);              /* fc2 ... iSTFT: 4 x (2,n) */
    int (*get_layer)(void *user, int l, float *state);        /* layer_floats values */
    int (*set_layer)(void *user, int l, const float *state);
    size_t layer_floats;
    void *user;
} umx_phased_backend;
typedef struct umx_p2p
{
    int (*send)(void *user, const float *buf, size_t n, int dst);
    int (*recv)(void *user, float *buf, size_t n, int src);
    void *user;
} umx_p2p;
/* out[4] (2,length) is written on rank 0 only (may be NULL elsewhere). */
int umx_split_inference_carry(const umx_phased_backend *be, const umx_p2p *p2p, int rank, int world, const float *audio,
                              int length, int segment_samples, float *const out[4], char *err);

/* ---- one track over several ranks, sharded by SOURCE MODEL as well (BASELINE north star: "the four source models ...
 * shard naturally").  world = G target groups x P pipeline stages, G = 4 / 2 / 1 as world is divisible by 4 / 2 / neither
 * (host/shard_plan.h).  The per-target loop of umx_inference (inference.cpp:70-186) is independent per target until
 * wiener_filter (inference.cpp:192-193): rank (g, p) runs the targets t % G == g of the segments s % P == p; the LSTM
 * state of a target travels only between the stages of its own group; of the G ranks that share a segment one (rotating
 * with the segment index) receives the other groups' target magnitudes, runs the Wiener filter + inverse STFT and sends
 * the weighted stems to rank 0, which adds them in segment order (umx.cpp:234-273).  G = 1 is
 * umx_split_inference_carry's schedule.  Bit-identical to umx_split_inference on one rank.
 * The transport must not block a send until the matching receive is posted (buffered or non-blocking sends: gloo isend,
 * MPI_Bsend / MPI_Isend): this single-threaded host form posts a rank's sends and receives in program order -- as does
 * umx_split_inference_carry, where a rendezvous send would deadlock two ranks that send to each other.  The device form
 * (host/mgpu.cpp) has no such requirement: it sends on a stream of its own. */
typedef struct umx_target_backend
{
    int (*begin)(void *user, const float *audio, int n, unsigned target_mask); /* front of a segment for the targets in the mask */
    int (*layer)(void *user, int l);                                   /* LSTM layer l of those targets */
    int (*get_state)(void *user, int l, int target, float *state);     /* target_layer_floats values: [2 dirs][h, c][hidden/2] */
    int (*set_state)(void *user, int l, int target, const float *state);
    int (*masks)(void *user);                                          /* fc2, fc3: the target magnitudes of those targets */
    int (*get_mag)(void *user, int target, float *mag);                /* mag_floats values, layout private to the backend */
    int (*set_mag)(void *user, int target, const float *mag);
    int (*finish)(void *user, float *const out[4]);                    /* Wiener + inverse STFT from all four magnitudes */
    int (*discard)(void *user);                                        /* this rank does not filter the segment */
    size_t target_layer_floats, mag_floats;
    void *user;
} umx_target_backend;
int umx_split_inference_targets(const umx_target_backend *be, const umx_p2p *p2p, int rank, int world, const float *audio,
                                int length, int segment_samples, float *const out[4], char *err);

/* Plan of a track: the (offset, length) of every segment split_inference will run (umx.cpp:214-218),
 * used by the multi-GPU scheduler.  Returns the number of segments; fills up to cap entries. */
int umx_segment_plan(int length, int segment_samples, int *offsets, int *lengths, int cap);
/* The triangular transition weight of sample k of a chunk (umx.cpp:197-206, 246). */
float umx_transition_weight(int k, int chunk_len, int segment_samples);

#ifdef __cplusplus
}
#endif
#endif /* UMX_HOST_H */
