/*
 * umx_oracle.h -- CPU restatement ("oracle") of the sevagh/umx.cpp segment-inference path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / reported CPU baseline.  The product path (umx.cpp_amd/) never links, loads or
 * calls it and has no CPU fallback.
 *
 * PARITY STATUS: "parity unpinned" against the real Eigen binary for everything after the
 * STFT.  The reference cannot be built in this project (Eigen, libnyquist, gtest are empty
 * git submodules) and its own tests only cover dsp (test/test_dsp.cpp).  This
 * restatement is pinned by (a) the reference's own dsp assertions restated in
 * tests/test_oracle_dsp.py (round trips <= 1e-4, 262,144-sample wav facts, 2049 bins),
 * (b) independent cross-checks generated in the build container and committed as fixtures
 * under tests/golden/ (numpy float64 STFT/iSTFT, torch.nn.LSTM / Linear / BatchNorm1d for the
 * network, a numpy restatement of the Wiener EM step), see tests/golden/make_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * repository root).  All arithmetic is fp32 unless a comment says otherwise.  Layouts at this
 * interface follow the reference: Eigen ColMajor, i.e. a (2,n) waveform is interleaved stereo,
 * spec(c,f,b) lives at c + 2*f + 2*T*b (complex = 2 floats re,im).
 */
#ifndef UMX_ORACLE_H
#define UMX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMXO_FFT 4096       /* dsp.hpp:17 FFT_WINDOW_SIZE */
#define UMXO_HOP 1024       /* dsp.hpp:19 FFT_HOP_SIZE    */
#define UMXO_BINS 2049      /* dsp.hpp:49 nb_bins         */
#define UMXO_CROP 1487      /* inference.cpp:55 nb_bins_cropped */
#define UMXO_TENSORS_PER_TARGET 43

/* Tensor order inside one target of the ggml file (the order the reference converter emits:
 * scripts/convert-umx-pth-to-ggml.py:127 iterates checkpoint.keys(); model.cpp:530-539 needs
 * bn3.running_var last). Index into the 43-entry table used by oracle_model_from_arrays(). */
const char *oracle_tensor_name(int idx);
/* number of floats of tensor idx for a given hidden size */
size_t oracle_tensor_numel(int idx, int hidden);

typedef struct oracle_model oracle_model;

/* model.cpp:42-574 restated: gunzip (zlib) -> parse -> dequantise.  NULL on any failure
 * (the reference returns false).  err (optional, >=256 bytes) receives a message. */
oracle_model *oracle_model_load(const char *path, char *err);
/* Build a model from 4*43 fp32 arrays in PyTorch layout (already dequantised). Copies. */
oracle_model *oracle_model_from_arrays(int hidden, const float *const *tensors);
void oracle_model_free(oracle_model *m);
int oracle_model_hidden(const oracle_model *m);
/* pointer to the dequantised fp32 data of (target, idx) in file (PyTorch row-major) order */
const float *oracle_model_tensor(const oracle_model *m, int target, int idx);

/* dsp.hpp:61-101 */
void oracle_hann_window(float *w /*4096*/);
void oracle_window_sumsq(int nb_frames, float *nw /*4096+1024*(nb_frames-1)*/);
int oracle_nb_frames(int n_samples_buf); /* dsp.hpp:48 */

/* dsp.cpp:141-176 (+109-128, 209-229).  audio: (2,n) interleaved, n <= n_buf.  The buffer is
 * sized for n_buf samples (stft_buffers(n_buf)), so T = n_buf/1024+1 whatever n is.
 * spec: ColMajor (2,T,2049) complex -> 2*2*T*2049 floats. */
void oracle_stft(const float *audio, int n, int n_buf, float *spec);
/* dsp.cpp:178-207, 231-258.  out: (2,n) interleaved */
void oracle_istft(const float *spec, int n, int n_buf, float *out);
/* raw fp32 real FFT pair used above (Eigen::FFT<float>, HalfSpectrum|Unscaled) */
void oracle_rfft4096(const float *in, float *out_complex /*2049*2*/);
void oracle_irfft4096(const float *in_complex /*2049*2*/, float *out);

/* lstm.hpp:10-20, lstm.cpp:41-99: persistent state, 3 layers x 2 dirs x {h,c} x (H/2).
 * Layout of the flat state: [target][layer][dir][0=h,1=c][H/2]. */
size_t oracle_stream_state_floats(int hidden); /* for all 4 targets */

/* lstm.cpp:101-179.  input (T x H) row-major; out (T x H) row-major; state = one target's
 * [3][2][2][H/2] block, updated in place. */
void oracle_lstm_forward(const oracle_model *m, int target, const float *input, int T,
                         float *state, float *out);

/* inference.cpp:70-186 for one target.  x: (T x 2974) row-major; mix_mag: ColMajor (2,T,2049).
 * Outputs (any may be NULL): fc1_out (T x H), lstm_out (T x H), mask (T x 4098) after the
 * output scale+relu, target_mag ColMajor (2,T,2049). */
void oracle_target_network(const oracle_model *m, int target, const float *x, const float *mix_mag,
                           int T, float *state, float *fc1_out, float *lstm_out, float *mask,
                           float *target_mag);
/* the same with the fc2/bn2/relu output (T x H, inference.cpp:127-140) as one more optional tap */
void oracle_target_network_ex(const oracle_model *m, int target, const float *x, const float *mix_mag,
                              int T, float *state, float *fc1_out, float *lstm_out, float *fc2_out,
                              float *mask, float *target_mag);

/* wiener.cpp:92-425.  mix_spec ColMajor (2,T,2049) complex (modified in place exactly like the
 * reference: divided by max_abs); target_mags: 4 x ColMajor (2,T,2049); y_out: 4 x complex. */
void oracle_wiener(float *mix_spec, const float *const *target_mags, int T, float *const *y_out);

/* inference.cpp:12-207.  audio (2,n) interleaved n<=n_buf; state: all 4 targets; out[4]: (2,n).
 * flags: bit0 = skip Wiener (config 2: y = target_mag * exp(i arg X)), bits 8..11 = target mask
 * to skip (a skipped target contributes an all-zero magnitude). Optional taps may be NULL. */
typedef struct
{
    float *spec;          /* (2,T,2049) complex ColMajor */
    float *mix_mag;       /* (2,T,2049) */
    float *x;             /* (T x 2974) row-major */
    float *fc1_out[4];    /* (T x H) */
    float *lstm_out[4];   /* (T x H) */
    float *mask[4];       /* (T x 4098) */
    float *target_mag[4]; /* (2,T,2049) ColMajor */
    float *y[4];          /* (2,T,2049) complex ColMajor, Wiener output */
    float *fc2_out[4];    /* (T x H), may be NULL */
} oracle_taps;
void oracle_umx_inference(const oracle_model *m, const float *audio, int n, int n_buf,
                          float *state, float *const *out, int flags, oracle_taps *taps);

/* umx.cpp:152-295 with sum_weight fully zero-initialised (the reference writes out
 * of bounds / leaves entries uninitialised for tracks < 29.6 s; the evident intent is zeros).
 * audio (2,length) -> out[4] (2,length).  segment_samples normally 2,646,000. */
void oracle_split_inference(const oracle_model *m, const float *audio, int length,
                            int segment_samples, float *const *out, int flags);
/* umx.cpp:99-150; offset is the value rand()%22050 took (4033 for the unseeded glibc rand()). */
void oracle_shift_inference(const oracle_model *m, const float *audio, int length,
                            int segment_samples, int offset, float *const *out, int flags);

int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
