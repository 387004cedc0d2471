/*
 * umx_hip.h -- C-ABI of the MI355X (gfx950) Open-Unmix segment-inference engine.
 *
 * Drop-in boundary for the reference's src/inference.cpp.  The reference has no plugin / FFI
 * surface; the seam is the C++ call
 *     std::vector<Eigen::MatrixXf> umxcpp::umx_inference(umx_model&, const Eigen::MatrixXf audio,
 *         stft_buffers, std::array<lstm_data,4>& streaming_lstm_data);      (src/inference.hpp:20-23)
 * made once per segment from umx.cpp:226-227, plus load_umx_model (src/model.hpp:58) that fills
 * the model.  Each entry point below names what it replaces.  Plain pointers and sizes only:
 * no Eigen, no torch types.  Every function returns an int status (UMX_OK == 0); the library
 * never calls exit() (the reference exits or returns false: dsp.cpp:27-44, model.cpp:59-64).
 *
 * A (2,n) Eigen ColMajor MatrixXf is exactly n interleaved stereo frames, so "audio" and "out"
 * buffers here are bit-identical to the reference's waveform matrices.
 *
 * Threading: one caller thread per context (as the reference: not re-entrant per stream state).
 */
#ifndef UMX_HIP_H
#define UMX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMX_OK 0
#define UMX_ERR_ARG 1     /* bad argument / shape (the reference: assert compiled out, or return false) */
#define UMX_ERR_HIP 2     /* a HIP runtime call failed; see umx_hip_last_error */
#define UMX_ERR_MODEL 3   /* missing / mis-shaped tensor (model.cpp:541-546, 582-591) */
#define UMX_ERR_TIMEOUT 4 /* persistent LSTM kernel gave up waiting (bounded spin) */
#define UMX_ERR_NODEVICE 5

#define UMX_DTYPE_F32 0
#define UMX_DTYPE_U8 1  /* model.cpp:578-619 load_single_matrix */
#define UMX_DTYPE_U16 2 /* model.cpp:623-665 load_single_matrix_uint16 */

/* flags of umx_hip_infer_segment* */
#define UMX_FLAG_NO_WIENER 0x1      /* BASELINE config 2: mix-phase estimate only (wiener.cpp:96-109) */
#define UMX_FLAG_SKIP_TARGET(t) (0x100 << (t)) /* BASELINE config 1 (vocals only = skip 0,1,2) */
#define UMX_FLAG_LSTM_STEPWISE 0x10 /* one launch per timestep instead of the persistent kernel */
#define UMX_FLAG_DEBUG_TAPS 0x20    /* keep what only the taps read: the filtered spectrograms for umx_hip_read_tap("y") (the fused kernel does not
                                     * write them otherwise) and, in track-batched contexts, the fp32 rows of the recurrence's layers ("lstm",
                                     * "lstm_l0", "lstm_l1": the batched kernels write the next GEMM's fp16 planes instead, csrc/lstm_batch.h) */
#define UMX_FLAG_LSTM_FORCE_SAFE 0x40 /* persistent kernel: never take the intra-XCD fast protocol */
#define UMX_FLAG_LSTM_PROFILE 0x80  /* persistent kernel: record per-phase cycle counters */
#define UMX_FLAG_DEBUG_LSTM_ABORT 0x2000 /* testing: the persistent LSTM launch of layer 1 gives up half way, exactly as if a
                                          hidden-state poll had timed out (exercises the recovery of umx_hip_sync) */
#define UMX_FLAG_RESET_SEGMENTS 0x4000 /* umx_hip_split_inference / _shift_inference / _separate_tracks with ONE track on a context made by
                                        * umx_hip_create_tracks: RESET MODE -- every segment starts from a zero lstm_data instead of the one its
                                        * predecessor left (umx.cpp:167-171 creates it once per track, :226-227 passes it to every segment): a
                                        * DECLARED deviation from the reference, opt-in.  The track's segments are then independent and ride as
                                        * the track lanes of one call (a 600 s track = 14 lanes of one pass).  Equals split_inference with
                                        * umx_lstm_set_zero (lstm.cpp:86-99) in front of every segment. */
#define UMX_FLAG_PRECISE_ACT 0x1000 /* LSTM gates with the device-library expf/tanhf + IEEE division instead of
                                       the hardware v_exp_f32 / v_rcp_f32 forms (~1e-7 abs difference) */

/* One tensor of the ggml-style weight file, as stored (scripts/convert-umx-pth-to-ggml.py:146-160
 * record = {scale, offset, n_dims, name_len, ne[], name, data}).  dtype F32 means `data` is already
 * dequantised fp32 in PyTorch row-major order and scale/offset are ignored. */
typedef struct umx_tensor_view
{
    const char *name; /* "fc1.weight", "lstm.weight_hh_l2_reverse", ... */
    int target;       /* 0 = bass, 1 = drums, 2 = other, 3 = vocals (convert script :104) */
    int dtype;        /* UMX_DTYPE_* */
    int n_dims;
    int ne[2];        /* PyTorch shape reversed, as in the file */
    float scale, offset;
    const void *data;
} umx_tensor_view;

typedef struct umx_hip_ctx umx_hip_ctx;

/* Replaces the result of load_umx_model (model.cpp:42-574) living on the device: dequantises
 * (q*scale+offset, model.cpp:610-616), re-lays the weights out for the kernels and uploads them.
 * segment_samples = the stft_buffers size (umx.cpp:160: 60 s * 44100 = 2,646,000): every segment,
 * also a shorter last one, is processed as T = segment_samples/1024+1 frames (dsp.cpp:214-217).
 * Needs 4 x 43 tensors (model.cpp:240-539 name dispatch).  hidden_size % 128 == 0. */
int umx_hip_create(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                   const umx_tensor_view *tensors, int n_tensors);
/* Weight residency (BASELINE config 5 / SURVEY 8(f)1).  Tensors handed over as u8 / u16 views (the ggml file's own
 * bytes: u8 fc1, W_ih, W_hh; u16 fc2, fc3 -- convert-umx-pth-to-ggml.py:127-160) stay that way in HBM by default
 * and q*scale+offset (model.cpp:610-616) is evaluated where they are consumed: in the GEMM's B-tile staging and in
 * the LSTM kernel's one-time W_hh register load.  139 MB of weights for UMX-L instead of 452 MB (fp32) / 630 MB
 * (three bf16 planes), bit-identical results, and measured FASTER (7.5 vs 8.0 ms per segment: the B operand moves
 * 1-2 bytes per weight instead of 6).  UMX_CREATE_DEQUANTISE_AT_LOAD (environment UMX_WEIGHTS_RESIDENT=expanded)
 * expands them at load time instead; fp32 views are always expanded.  UMX_CREATE_QUANTISED_RESIDENT is accepted
 * for compatibility and is the default. */
#define UMX_CREATE_QUANTISED_RESIDENT 0x1u
#define UMX_CREATE_DEQUANTISE_AT_LOAD 0x8u
/* The dense stack (fc1, W_ih, fc2, fc3 -- inference.cpp:86,127,143, lstm.cpp:132-135) runs on the 16-bit matrix cores
 * with split operands and fp32 accumulation (csrc/gemm_planes.h, csrc/gemm_bf16x3.h, below): the dropped terms are below
 * 2^-22 .. 2^-26 of a product, and against a float64 evaluation the result is as close as an fp32-MFMA kernel's (measured:
 * slightly closer; profiles/r02_accuracy_vs_float64.txt).  Rounds 1-2 also carried that fp32-MFMA flavour
 * (UMX_CREATE_GEMM_F32): removed in round 3; the flag is now refused with UMX_ERR_ARG. */
#define UMX_CREATE_GEMM_F32 0x4u
/* u8-resident weights (fc1, W_ih; W_hh in the batched LSTM kernel) on the bf16 matrix cores: q - 128 is an integer in
 * [-128, 127] and EXACT in bf16 and fp16, so by default the weight is ONE term (three products with the bf16-split
 * activation instead of six; two with the fp16 planes of csrc/gemm_planes.h) and the affine map of model.cpp:610-616 is applied to the accumulated sum:
 *     sum_k a_k (q_k s + o) = s sum_k a_k (q_k - 128) + (o + 128 s) sum_k a_k.
 * The reference rounds q*s+o to fp32 per weight first; the two differ by exactly that rounding (~1e-7 of the dot
 * product, the size of one fp32 rounding of the sum).  UMX_CREATE_U8_DEQUANT (environment UMX_U8=dequant) keeps the
 * per-weight form (dequantise, split in three, six products): bit-identical to UMX_CREATE_DEQUANTISE_AT_LOAD. */
#define UMX_CREATE_U8_DEQUANT 0x20u
/* The split-operand GEMM flavour exists in two forms.  csrc/gemm_planes.h (default of track-batched contexts): every
 * activation matrix is split once, by a small kernel, into TWO fp16 planes of the row scaled by a power of two (+ row
 * sums and inverse scales), every weight matrix is re-encoded once at load time as exact fp16 integer planes (u8: 1,
 * u16: 2; fp32: 2 split terms), and the GEMM itself only moves 256 x 256 tiles global -> LDS by DMA, over all track lanes
 * at once, and issues matrix-core instructions (2 products per fp32 product for u8 weights, 4 for u16).
 * csrc/gemm_bf16x3.h (default of single-track contexts): keeps u8 / u16 weights as stored and splits both operands while
 * staging every 128 x 128 tile; small enough in registers and LDS to share the CUs with the two co-resident single-track
 * LSTM grids of the latency pipeline.  UMX_CREATE_GEMM_STAGED / UMX_CREATE_GEMM_PLANES (environment UMX_GEMM=bf16x3 /
 * planes) force one or the other. */
#define UMX_CREATE_GEMM_STAGED 0x40u
#define UMX_CREATE_GEMM_PLANES 0x80u
int umx_hip_create_ex(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                      const umx_tensor_view *tensors, int n_tensors, unsigned create_flags);
/* Track batching.  The reference is one track per process (umx.cpp:26-97) and its LSTM is one
 * matrix-vector product per step (lstm.cpp:132-161) -- on a GPU that step is bound by the cross-CU hand-off latency,
 * not by arithmetic.  A context created for n_tracks (1..UMX_MAX_TRACKS = 64) independent tracks holds that many "track lanes", each
 * with its own streaming LSTM state (= its own std::array<lstm_data,4>, umx.cpp:167-171) and activation buffers;
 * umx_hip_infer_batch* runs one segment of every lane per call, and the recurrence of all lanes is ONE launch per
 * layer in which W_hh.h is a matrix-matrix product on the matrix cores (u8 W_hh as one exact fp16 plane against two fp16
 * planes of h, fp32 accumulate: csrc/lstm_batch.h).  The LSTM flavour is fixed per context: n_tracks > 1 (or UMX_CREATE_LSTM_BATCHED, environment
 * UMX_LSTM=batched, on a 1-track context) selects the batched kernel for every call, so a track's result never
 * depends on how many lanes a call uses or which lane it sits in (bitwise; tests/test_gpu_batch.py).  Against the
 * single-track kernel the results agree to fp32 rounding (different summation order), not bitwise.
 * Hidden 1024 (UMX-L) and 512 (umxhq) with the u8-resident W_hh of a quantised model, any number of lanes: csrc/lstm_batch8.h -- a
 * workgroup serves an octet of 8 lanes x 64 hidden units (N of the matrix instruction = 8 lanes x the 2 planes of h); hidden 1024: 32
 * lanes per launch with one octet per workgroup, 33 .. 64 with two octets per workgroup IN TURN (one octet's hand-off travels under
 * the other's matrix and gate phases); hidden 512: its 64 lanes side by side.  Other hidden sizes / fp32-resident W_hh:
 * lstm_batch_kernel of csrc/lstm_batch.h, a group of 16 lanes per launch, the groups of a larger context one launch after the other. */
#define UMX_CREATE_LSTM_BATCHED 0x10u
#define UMX_MAX_TRACKS 64
int umx_hip_create_tracks(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                          const umx_tensor_view *tensors, int n_tensors, unsigned create_flags, int n_tracks);
int umx_hip_n_tracks(const umx_hip_ctx *ctx);
/* Segments the context keeps in flight (= its pipeline slots, each with its own stream): 2.  The buffers of that many
 * CONSECUTIVE asynchronous calls must be distinct (ordering contract of umx_hip_infer_segment_device below). */
int umx_hip_pipeline_depth(const umx_hip_ctx *ctx);
int umx_hip_lstm_is_batched(const umx_hip_ctx *ctx);
size_t umx_hip_weight_bytes(const umx_hip_ctx *ctx); /* HBM bytes held by the model's weight matrices */
void umx_hip_destroy(umx_hip_ctx *ctx);
const char *umx_hip_last_error(const umx_hip_ctx *ctx); /* never NULL; also valid for ctx == NULL (create errors) */

/* Streaming LSTM state = the h/c members of lstm_data (lstm.hpp:10-16), created zeroed once per
 * track (umx.cpp:167-171, lstm.cpp:82) and carried across segments.
 * Layout: [4 targets][3 layers][2 dirs][2: h, c][hidden/2] floats. */
size_t umx_hip_stream_floats(const umx_hip_ctx *ctx);
int umx_hip_stream_reset(umx_hip_ctx *ctx);                 /* create_lstm_data / umx_lstm_set_zero */
int umx_hip_stream_get(umx_hip_ctx *ctx, float *host_dst);  /* checkpoint / multi-GPU hand-off */
int umx_hip_stream_set(umx_hip_ctx *ctx, const float *host_src);
/* The same per track lane (the three above address lane 0); reset with track < 0 clears every lane. */
int umx_hip_track_stream_reset(umx_hip_ctx *ctx, int track);
int umx_hip_track_stream_get(umx_hip_ctx *ctx, int track, float *host_dst);
int umx_hip_track_stream_set(umx_hip_ctx *ctx, int track, const float *host_src);

/* umx_inference (inference.cpp:12-207): one segment -> 4 stems.
 * audio: n interleaved stereo frames (2,n), 1 <= n <= segment_samples.  out[t]: (2,n) each.
 * Host-pointer form: H2D + kernels + D2H, synchronous (umx_inference returns its outputs). */
int umx_hip_infer_segment(umx_hip_ctx *ctx, const float *audio_host, int n, float *const out_host[4],
                          unsigned flags);
/* The same without the final wait: H2D and kernels are queued on the stream of the pipeline slot the segment runs in
 * (consecutive calls go round the slots), the D2H on a copy stream behind them.  With PINNED host buffers, and distinct
 * buffers for umx_hip_pipeline_depth() consecutive calls, one segment's transfers overlap the others' kernels.  Results
 * are valid after umx_hip_sync.
 * BUFFER CONTRACT: audio_host and out_host of every call queued since the last umx_hip_sync must stay valid and
 * unchanged until that sync returns -- the timeout recovery described there uploads the audio again. */
int umx_hip_infer_segment_async(umx_hip_ctx *ctx, const float *audio_host, int n, float *const out_host[4],
                                unsigned flags);
/* Device-pointer form: buffers already in HBM (audio 2*n floats, out[t] 2*n floats each); asynchronous.
 * ORDERING CONTRACT: consecutive calls are queued round-robin on umx_hip_pipeline_depth() internal streams (the
 * cross-segment pipeline), so (1) audio_dev and out_dev must stay untouched by the caller until umx_hip_sync -- or until
 * the caller's stream has been ordered behind the engine with umx_hip_order_before; (2) umx_hip_pipeline_depth()
 * CONSECUTIVE calls must be given DISTINCT out_dev buffers (that many segments are in flight together); (3) work the caller queued on a stream
 * of its own that produces audio_dev is ordered in front of the next call with umx_hip_order_after.
 * A caller that fences with umx_hip_order_before (and then recycles its buffers) gives up the timeout recovery for the
 * calls queued so far: a persistent-kernel timeout among them is reported as UMX_ERR_TIMEOUT by the next umx_hip_sync.
 * Track-batched contexts (umx_hip_create_tracks with more than one lane) run the KERNELS of consecutive calls one after the
 * other (their workgroups take whole CUs: side by side they only wait for each other); the slots' streams still let a
 * call's uploads and downloads run beside another call's kernels, and the contract above is unchanged. */
int umx_hip_infer_segment_device(umx_hip_ctx *ctx, const float *audio_dev, int n, float *const out_dev[4],
                                 unsigned flags);
/* One segment of each of n_tracks track lanes (lane i = the i-th track of the context, its LSTM state carries from
 * call to call): audio[i] = (2,n[i]) interleaved or NULL for a lane that sits this call out (its state is kept),
 * out[4*i + t] = stem t of track i.  Same three forms and the same ordering contract as above. */
int umx_hip_infer_batch(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *n,
                        float *const *out_host, unsigned flags);
int umx_hip_infer_batch_async(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *n,
                              float *const *out_host, unsigned flags);
int umx_hip_infer_batch_device(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_dev, const int *n,
                               float *const *out_dev, unsigned flags);
/* Waits for everything queued.  A persistent LSTM launch needs its whole grid co-resident; inside a process that is
 * guaranteed (launches of all contexts on a device pass one admission gate), but another PROCESS on the same GPU can
 * still occupy the CUs, in which case the launch gives up after a bounded spin instead of hanging.  umx_hip_sync then
 * RECOVERS: the stream state of every lane is restored to what it was before the first segment queued since the last
 * sync (a per-layer copy is kept for up to 8 queued calls), those segments are run again with the per-step driver
 * (bit-identical), and later calls of this context use that driver.  UMX_OK after a recovery; umx_hip_last_error then
 * starts with "recovered".  UMX_ERR_TIMEOUT (streaming state reset to zero) only when recovery is impossible: more
 * than 8 calls were queued since the last sync, or umx_hip_order_before released the caller's buffers of those calls.
 * The stream-state entry points (umx_hip_[track_]stream_{reset,get,set}) synchronise through this function, so a
 * sequence infer(A); stream_set; infer(B) replays A before the state is changed and B after it, never across it. */
int umx_hip_sync(umx_hip_ctx *ctx);
/* hip_stream = a hipStream_t of the caller.  order_after: everything queued by LATER calls on this context starts
 * only after what is on hip_stream now.  order_before: hip_stream waits for everything queued on the context so far. */
int umx_hip_order_after(umx_hip_ctx *ctx, void *hip_stream);
int umx_hip_order_before(umx_hip_ctx *ctx, void *hip_stream);
void *umx_hip_stream_handle(umx_hip_ctx *ctx); /* the internal hipStream_t the most recent segment was queued on */

/* The whole track on the device: shift_inference (umx.cpp:99-150) around split_inference (umx.cpp:152-295).
 * One upload of the (2,length) interleaved track, the 60 s segments (stride 0.75 * segment_samples,
 * umx.cpp:181) queued back to back so that consecutive segments overlap in the engine's two pipeline slots,
 * the triangular-weight overlap-add (umx.cpp:197-260) and the division by the weight sum (umx.cpp:264-273)
 * in HBM, one download of the 4 stems.  Resets the streaming state first (umx.cpp:167-171).  Bit-identical to
 * umx_split_inference / umx_shift_inference of umx_host.h driving umx_hip_infer_segment; sum_weight is zeroed
 * (SURVEY F4).  offset: samples of leading silence inside the MAX_SHIFT buffer (umx.cpp:115); < 0 = the
 * reference's unseeded rand() % 22050 = UMX_REFERENCE_SHIFT.  progress (may be NULL) is called once per QUEUED segment. */
#define UMX_MAX_SHIFT 22050 /* inference.hpp:14 MAX_SHIFT_SECS * 44100 */
/* umx.cpp:115 draws rand() % 22050 and the reference never seeds rand(): with glibc every run of its CLI shifts by these
 * 4033 samples.  "offset < 0" below means THIS value, not a call of rand() here -- in a process that links the HIP runtime
 * (or anything else that draws from rand() first) the first rand() is no longer the reference's (found by a test, round 3). */
#define UMX_REFERENCE_SHIFT 4033
int umx_hip_split_inference(umx_hip_ctx *ctx, const float *audio_host, int length, float *const out_host[4],
                            unsigned flags, void (*progress)(float, void *), void *progress_user);
int umx_hip_shift_inference(umx_hip_ctx *ctx, const float *audio_host, int length, int offset, float *const out_host[4],
                            unsigned flags, void (*progress)(float, void *), void *progress_user);

/* The same for n_tracks tracks at once, one per track lane of a context made by umx_hip_create_tracks: call s of the
 * segment loop runs segment s of EVERY track that still has one (their LSTM recurrences share one launch per layer),
 * a finished track's lane sits idle.  audio_host[i] (2,length[i]), out_host[4*i + t] (2,length[i]); shift_offset[i] < 0:
 * split_inference, >= 0: shift_inference with that offset.  Track i's stems are bit-identical to running it alone on a
 * context with the batched LSTM kernel. */
int umx_hip_separate_tracks(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *length,
                            const int *shift_offset, float *const *out_host, unsigned flags, void (*progress)(float, void *),
                            void *progress_user);

/* One segment phase by phase: front (STFT, fc1, W_ih layer 0) | LSTM layer 0 | 1 | 2 | back (fc2, fc3, Wiener,
 * iSTFT).  Same kernels and results as umx_hip_infer_segment; the cuts are where the reference's per-chain
 * (h, c) (lstm.cpp:116-161: read at the start of a layer, left behind at its end) crosses from the GPU that ran
 * the previous segment (SURVEY 8e "carry" mode): set layer l's state, run layer l, get it, pass it on.
 * Layer state layout: [4 targets][2 dirs][2: h, c][hidden/2] floats. */
size_t umx_hip_stream_layer_floats(const umx_hip_ctx *ctx);
int umx_hip_stream_get_layer(umx_hip_ctx *ctx, int layer, float *host_dst);
int umx_hip_stream_set_layer(umx_hip_ctx *ctx, int layer, const float *host_src);
int umx_hip_segment_begin(umx_hip_ctx *ctx, const float *audio_host, int n, unsigned flags);
int umx_hip_segment_lstm_layer(umx_hip_ctx *ctx, int layer); /* 0, 1, 2 in order */
int umx_hip_segment_end(umx_hip_ctx *ctx, float *const out_host[4]);

/* The same cut points for a driver that keeps everything in HBM and on ONE stream (host/mgpu.cpp: RCCL send / recv
 * of the state and of the weighted stems on device pointers, no host bounce).  None of these waits for the device;
 * all work is queued on umx_hip_phase_stream in call order, so a collective queued on that stream between two of
 * them is ordered against the kernels on both sides.
 *   umx_hip_segment_begin_device   front of a segment, audio already in HBM
 *   umx_hip_segment_lstm_layer     (as above)
 *   umx_hip_segment_end_device     back of the segment into 4 device buffers (2,n)
 *   umx_hip_stream_state_device    device address of track lane 0's stream state, [4 targets][3 layers][2 dirs][2: h, c][hidden/2]:
 *                                  layer l of target t is the 4 * hidden/2 floats at ((t * 3 + l) * 4) * hidden/2
 *   umx_hip_weight_stems_device    stems[t][k] *= transition weight of sample k (umx.cpp:197-206, 246), in place
 *   umx_hip_track_accumulate_device / _normalise_device   umx.cpp:234-273 on device buffers: track (2,length) x 4 +=
 *                                  already weighted stems at `offset`, sum_weight += weights; then track /= sum_weight */
/* The back of a segment in two halves, for the driver that shards a track by SOURCE MODEL (host/mgpu.cpp, target mode):
 * the per-target loop of umx_inference (inference.cpp:70-186) is independent per target until wiener_filter
 * (inference.cpp:192-193), so a GPU runs begin / lstm_layer x 3 / masks with the other targets skipped
 * (UMX_FLAG_SKIP_TARGET), the target magnitudes travel to the GPU that filters this segment, and that GPU finishes:
 *   umx_hip_segment_masks_device   fc2, fc3, mask x |X| of the targets that are not skipped (after layer 2)
 *   umx_hip_target_mag_device      device address of target t's MASK planes [2][T][2176] (2049 bins + padding per row;
 *                                  *floats = the size) of the phased segment -- the target magnitude is mask x |X|
 *                                  (inference.cpp:175-183), which _finish_device forms from its own spectrogram: read it
 *                                  after _masks_device, or write a peer's result there before _finish_device
 *   umx_hip_segment_finish_device  Wiener EM (or the mixture phase), inverse STFT, overlap-add from ALL four magnitude
 *                                  buffers as they are (a skipped target is NOT zero-filled here) into 4 device buffers
 * umx_hip_segment_end_device == _masks_device, zero-fill of the skipped targets, _finish_device. */
int umx_hip_segment_masks_device(umx_hip_ctx *ctx);
float *umx_hip_target_mag_device(umx_hip_ctx *ctx, int target, size_t *floats);
int umx_hip_segment_finish_device(umx_hip_ctx *ctx, float *const out_dev[4]);
int umx_hip_segment_discard(umx_hip_ctx *ctx); /* closes the phased segment where it stands (a rank that only contributes magnitudes) */
/* Persistent LSTM launches of every context on `device` leave `cus` compute units out of their co-residency budget
 * (the admission gate of umx_hip_sync's comment): room for kernels that are not this engine's and that may sit on a CU
 * for a long time -- RCCL's send / recv kernels in host/mgpu.cpp.  Process-wide, per device: cus > 0 files a request (the SUM of
 * the outstanding ones applies, at most half the chip: every driver keeps its own transfer kernels resident), -cus gives one
 * request of that size back (UMX_ERR_ARG if there is none), 0 drops every request (tests only). */
int umx_hip_gate_reserve(int device, int cus);
void *umx_hip_phase_stream(umx_hip_ctx *ctx);
float *umx_hip_stream_state_device(umx_hip_ctx *ctx);
int umx_hip_segment_begin_device(umx_hip_ctx *ctx, const float *audio_dev, int n, unsigned flags);
int umx_hip_segment_end_device(umx_hip_ctx *ctx, float *const out_dev[4]);
int umx_hip_weight_stems_device(umx_hip_ctx *ctx, float *const stems_dev[4], int n, void *hip_stream);
int umx_hip_track_accumulate_device(umx_hip_ctx *ctx, float *const track_dev[4], float *sum_weight_dev,
                                    const float *const weighted_dev[4], int offset, int n, void *hip_stream);
int umx_hip_track_normalise_device(umx_hip_ctx *ctx, float *const track_dev[4], const float *sum_weight_dev, int length,
                                   void *hip_stream);

/* Geometry */
int umx_hip_nb_frames(const umx_hip_ctx *ctx);       /* T = segment_samples/1024 + 1 (dsp.hpp:48) */
int umx_hip_segment_samples(const umx_hip_ctx *ctx);
int umx_hip_hidden(const umx_hip_ctx *ctx);

/* Stage taps for parity tests (D2H copy of an intermediate of the LAST inferred segment).
 * what: "spec" [2][T][2049] complex | "mix_mag" [2][T][2049] | "x" [T][2976] |
 *       "fc1" [T][H] | "lstm" [T][H] | "fc2" [T][H] | "mask" [T][4098] |
 *       "target_mag" [2][T][2049] | "y" [2][T][2049] complex (needs UMX_FLAG_DEBUG_TAPS; so does "lstm" in a track-batched
 *       context) | "max_abs" [1];
 *       ("mix_mag", "target_mag" and "mask" are not buffers of the engine any more -- fc3 writes the mask in a padded
 *       layout and the Wiener kernels form mask x |X| in registers -- a tap kernel computes them on demand with the
 *       device functions the hot kernels use, so they hold the bits the pipeline works with)
 *       suffix "#k" selects track lane k (default 0), "@s" pipeline slot s (default: the most recent)
 * Returns the number of floats written (or needed when dst == NULL), < 0 on error. */
long umx_hip_read_tap(umx_hip_ctx *ctx, const char *what, int target, float *dst, size_t capacity_floats);

/* Per-stage device time of the LAST segment, measured with hipEvents on the context's stream.
 * names/ms are filled up to `cap` entries; returns the number of stages. */
int umx_hip_stage_times(umx_hip_ctx *ctx, const char **names, float *ms, int cap);
/* Same for pipeline slot 0 or 1 (consecutive segments alternate slots; when two segments were queued
 * back to back their spans overlap, so a stage's time includes interference from the other slot). */
int umx_hip_stage_times_slot(umx_hip_ctx *ctx, int slot_index, const char **names, float *ms, int cap);
/* The same stages with the MAIN kernel alone where a stage launches a preparing kernel first (the GEMM stages of plane
 * contexts: split_planes_kernel, then the GEMM): the time from an event recorded between the two to the next stage's
 * event.  Stages without such a kernel report their stage time.  slot_index < 0: the slot of the last call. */
int umx_hip_stage_kernel_times_slot(umx_hip_ctx *ctx, int slot_index, float *ms, int cap);
/* 1 if the LSTM layers of the last segment ran in the persistent (one launch per layer) kernel,
 * 0 if the per-timestep driver was used (flag, unsupported hidden size, or grid not co-resident). */
int umx_hip_lstm_was_persistent(const umx_hip_ctx *ctx);
/* The recurrence kernel the last LSTM layer launch used: "lstm_persistent_kernel" / "lstm_step_kernel" (one track),
 * "lstm_batch8_kernel" (track-batched, hidden 1024 / 512, u8-resident W_hh), "lstm_batch_kernel" (every other track-batched context). */
const char *umx_hip_lstm_kernel_name(const umx_hip_ctx *ctx);
/* The GEMM kernel the last call launched for stage `mode` (0 fc1, 1 W_ih, 2 fc2, 3 fc3): "gemm_planes_ps_kernel" (persistent walk),
   "gemm_planes_pp_kernel", "gemm_planes_kernel", "gemm_bf16x3_kernel" or "none".  For bench.py's per-kernel figures: no reference analogue. */
const char *umx_hip_gemm_kernel_name(const umx_hip_ctx *ctx, int mode);
/* 0 = per-timestep driver, 1 = persistent kernel with the placement-independent (sc1) hand-off,
 * 2 = persistent kernel whose census found every chain on one XCD (intra-L2 hand-off); < 0 on error */
int umx_hip_lstm_mode(umx_hip_ctx *ctx);
/* UMX_FLAG_LSTM_PROFILE: shader-clock cycles summed over the T steps of each layer, for waves 0 and 1
 * of workgroup (chain 0, slice 0): out48[(layer*2 + wave)*8 + {0 poll, 1 dot, 2 barrier, 3 gates, 4 steps}] */
/* Testing: the per-bin arithmetic both Wiener filter kernels share (|X|, PSDs with F5, Cxx with F6, closed-form inverse, y_j =
 * v_j R_j (Cxx^-1 x) * max_abs; wiener.cpp:301-400) on caller-given bins: X [n][2][re, im], masks [n][4][2], R [n][4][R00, Re R01,
 * Im R01, R11], y [n][4][2][re, im]; max_abs >= 1 (wiener.cpp:51).  Host pointers; needs a current HIP device. */
int umx_hip_debug_wiener_bins(int n, const float *X, const float *masks, const float *R, float max_abs, float *y);
int umx_hip_debug_lstm_profile(umx_hip_ctx *ctx, unsigned long long *out48);
/* Where the workgroups of the last profiled one-track recurrence launch ran (UMX_FLAG_LSTM_PROFILE): out[i] for workgroup i =
 * XCC id << 48 | chain << 40 | slice << 32 | the 32-bit HW_ID register (CU, shader array, shader engine).  n <= 512. */
int umx_hip_debug_lstm_placement(umx_hip_ctx *ctx, unsigned long long *out, int n);
/* Debugging (tools/bx_guard.py): queue `launches` guard kernels on a private stream -- each workgroup keeps a 36 KB
 * pattern in LDS and re-verifies it `rounds` times -- beside whatever the caller queues next; launches == 0 waits
 * for them and returns {words found changed, events} in out2.  Used to show that no kernel of this engine writes
 * into another workgroup's LDS. */
int umx_hip_debug_lds_guard(umx_hip_ctx *ctx, int launches, int rounds, unsigned *out2);
/* Testing hook (no GPU needed): the host-side fp32 -> fp16 conversion (round to nearest even, subnormals, overflow to
 * infinity) with which weights are re-encoded as fp16 planes at load time (csrc/gemm_planes.h); returns the 16 bits. */
unsigned umx_hip_debug_f16_bits(float x);

#ifdef __cplusplus
}
#endif
#endif /* UMX_HIP_H */
