/*
 * umx_oracle.cpp -- CPU restatement of the sevagh/umx.cpp hot path.  See umx_oracle.h:
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" vs the real Eigen binary after the STFT.
 *
 * Written from the behaviour of the reference (file:line cited per function); no reference
 * source is copied.  Buffers are flat fp32; "CM" below means Eigen ColMajor indexing.
 */
#include "umx_oracle.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{

typedef std::complex<float> cf;

const int NFFT = UMXO_FFT, HOP = UMXO_HOP, NB = UMXO_BINS, CROP = UMXO_CROP;
const int NIN = 2 * UMXO_CROP;  // 2974  inference.cpp:41
const int NOUT = 2 * UMXO_BINS; // 4098  inference.cpp:53

int g_threads = 0;
thread_local int tl_inner = 0; /* thread budget inside an outer parallel-over-targets region */
int nthreads_total()
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
int nthreads()
{
    if (tl_inner > 0)
        return tl_inner;
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- tensor table (a15) */
enum
{
    T_INPUT_MEAN = 0,
    T_INPUT_SCALE,
    T_OUTPUT_SCALE,
    T_OUTPUT_MEAN,
    T_FC1_W,
    T_BN1_W,
    T_BN1_B,
    T_BN1_RM,
    T_BN1_RV,
    T_LSTM0, /* 24 entries: layer l, dir d: base = T_LSTM0 + (l*2+d)*4 : ih_w, hh_w, ih_b, hh_b */
    T_FC2_W = T_LSTM0 + 24,
    T_BN2_W,
    T_BN2_B,
    T_BN2_RM,
    T_BN2_RV,
    T_FC3_W,
    T_BN3_W,
    T_BN3_B,
    T_BN3_RM,
    T_BN3_RV,
    T_COUNT
};
static_assert(T_COUNT == UMXO_TENSORS_PER_TARGET, "43 tensors per target");

std::string tensor_name(int idx)
{
    static const char *fixed_head[] = {"input_mean", "input_scale",      "output_scale",
                                       "output_mean", "fc1.weight",      "bn1.weight",
                                       "bn1.bias",    "bn1.running_mean", "bn1.running_var"};
    static const char *fixed_tail[] = {"fc2.weight",       "bn2.weight",      "bn2.bias",
                                       "bn2.running_mean", "bn2.running_var", "fc3.weight",
                                       "bn3.weight",       "bn3.bias",        "bn3.running_mean",
                                       "bn3.running_var"};
    if (idx < T_LSTM0)
        return fixed_head[idx];
    if (idx >= T_FC2_W)
        return fixed_tail[idx - T_FC2_W];
    int k = idx - T_LSTM0;
    int layer = k / 8, dir = (k / 4) % 2, which = k % 4;
    static const char *w[] = {"weight_ih", "weight_hh", "bias_ih", "bias_hh"};
    std::string s = std::string("lstm.") + w[which] + "_l" + std::to_string(layer);
    if (dir)
        s += "_reverse";
    return s;
}

/* PyTorch shape (rows=out, cols=in) of tensor idx; 1-D tensors have cols = 0 */
void tensor_shape(int idx, int H, int *rows, int *cols)
{
    int Hl = H / 2, G = 2 * H; /* 4*Hl gates */
    *cols = 0;
    switch (idx)
    {
    case T_INPUT_MEAN:
    case T_INPUT_SCALE:
        *rows = CROP;
        return;
    case T_OUTPUT_SCALE:
    case T_OUTPUT_MEAN:
        *rows = NB;
        return;
    case T_FC1_W:
        *rows = H;
        *cols = NIN;
        return;
    case T_FC2_W:
        *rows = H;
        *cols = 2 * H;
        return;
    case T_FC3_W:
        *rows = NOUT;
        *cols = H;
        return;
    case T_BN1_W:
    case T_BN1_B:
    case T_BN1_RM:
    case T_BN1_RV:
    case T_BN2_W:
    case T_BN2_B:
    case T_BN2_RM:
    case T_BN2_RV:
        *rows = H;
        return;
    case T_BN3_W:
    case T_BN3_B:
    case T_BN3_RM:
    case T_BN3_RV:
        *rows = NOUT;
        return;
    default:
        break;
    }
    int which = (idx - T_LSTM0) % 4;
    *rows = G;
    if (which == 0)
        *cols = H;
    else if (which == 1)
        *cols = Hl;
}

size_t tensor_numel(int idx, int H)
{
    int r, c;
    tensor_shape(idx, H, &r, &c);
    return (size_t)r * (size_t)(c ? c : 1);
}

/* convert-umx-pth-to-ggml.py:146: u16 for names containing bn2, bn3, fc2, fc3; else u8.
 * model.cpp dispatches the same split by name (load_single_matrix vs _uint16). */
bool tensor_is_u16(const std::string &name)
{
    return name.find("bn2") != std::string::npos || name.find("bn3") != std::string::npos ||
           name.find("fc2") != std::string::npos || name.find("fc3") != std::string::npos;
}

} // namespace

struct oracle_model
{
    int hidden;
    std::vector<float> t[4][UMXO_TENSORS_PER_TARGET];
    const float *p(int target, int idx) const { return t[target][idx].data(); }
    const float *lstm(int target, int layer, int dir, int which) const
    {
        return t[target][T_LSTM0 + (layer * 2 + dir) * 4 + which].data();
    }
};

extern "C" const char *oracle_tensor_name(int idx)
{
    static std::string names[UMXO_TENSORS_PER_TARGET];
    if (idx < 0 || idx >= UMXO_TENSORS_PER_TARGET)
        return "";
    if (names[idx].empty())
        names[idx] = tensor_name(idx);
    return names[idx].c_str();
}

extern "C" size_t oracle_tensor_numel(int idx, int hidden) { return tensor_numel(idx, hidden); }

extern "C" oracle_model *oracle_model_from_arrays(int hidden, const float *const *tensors)
{
    oracle_model *m = new oracle_model;
    m->hidden = hidden;
    for (int tg = 0; tg < 4; ++tg)
        for (int i = 0; i < T_COUNT; ++i)
        {
            size_t n = tensor_numel(i, hidden);
            const float *src = tensors[tg * T_COUNT + i];
            m->t[tg][i].assign(src, src + n);
        }
    return m;
}

extern "C" void oracle_model_free(oracle_model *m) { delete m; }
extern "C" int oracle_model_hidden(const oracle_model *m) { return m->hidden; }
extern "C" const float *oracle_model_tensor(const oracle_model *m, int target, int idx)
{
    return m->p(target, idx);
}

/* model.cpp:42-574 restated.  Differences, all deliberate and documented in DESIGN.md:
 * decompresses to memory instead of ./temp.decompressed (model.cpp:56-84), accepts a
 * non-gzipped file too (gzread passes plain files through), never exits. */
extern "C" oracle_model *oracle_model_load(const char *path, char *err)
{
    auto fail = [&](const char *msg) -> oracle_model * {
        if (err)
            snprintf(err, 256, "%s", msg);
        return nullptr;
    };
    gzFile gz = gzopen(path, "rb"); /* model.cpp:58 */
    if (!gz)
        return fail("failed to open model file");
    std::vector<unsigned char> buf;
    {
        unsigned char chunk[1 << 16];
        int n;
        while ((n = gzread(gz, chunk, sizeof chunk)) > 0)
            buf.insert(buf.end(), chunk, chunk + n);
        gzclose(gz);
    }
    size_t pos = 0;
    auto rd = [&](void *dst, size_t n) -> bool {
        if (pos + n > buf.size())
            return false;
        memcpy(dst, buf.data() + pos, n);
        pos += n;
        return true;
    };
    uint32_t magic = 0, hidden = 0;
    if (!rd(&magic, 4) || magic != 0x756d7867u) /* model.cpp:101-106 */
        return fail("invalid model data (bad magic)");
    if (!rd(&hidden, 4)) /* model.cpp:109 */
        return fail("truncated header");
    oracle_model *m = new oracle_model;
    m->hidden = (int)hidden;
    int target = 0;
    int n_loaded = 0;
    for (;;)
    {
        float scale, offset;
        int32_t n_dims, name_len;
        if (!rd(&scale, 4)) /* EOF between records: model.cpp:228-232 */
            break;
        if (!rd(&offset, 4) || !rd(&n_dims, 4) || !rd(&name_len, 4) || n_dims < 0 || n_dims > 2 ||
            name_len < 0 || name_len > 256)
        {
            delete m;
            return fail("truncated or corrupt tensor header");
        }
        int32_t ne[2] = {1, 1};
        size_t nel = 1;
        for (int i = 0; i < n_dims; ++i)
        {
            if (!rd(&ne[i], 4))
            {
                delete m;
                return fail("truncated dims");
            }
            nel *= (size_t)ne[i];
        }
        std::string name(name_len, '\0');
        if (!rd(&name[0], name_len))
        {
            delete m;
            return fail("truncated name");
        }
        if (target >= 4)
        {
            delete m;
            return fail("more than 4 targets in file");
        }
        int idx = -1;
        for (int i = 0; i < T_COUNT; ++i)
            if (tensor_name(i) == name)
                idx = i;
        if (idx < 0) /* model.cpp:541-546: unknown name -> loaded_size 0 -> false */
        {
            delete m;
            return fail(("failed to load " + name).c_str());
        }
        int rows, cols;
        tensor_shape(idx, (int)hidden, &rows, &cols);
        /* file dims are the PyTorch shape reversed (convert script :154-155); the loader
         * compares against its ColMajor (ne0, ne1) matrices: model.cpp:582-591 */
        int want0 = cols ? cols : rows, want1 = cols ? rows : 1;
        if ((size_t)want0 * want1 != nel || ne[0] != want0 || ne[1] != want1)
        {
            delete m;
            return fail(("tensor '" + name + "' has wrong size in model file").c_str());
        }
        std::vector<float> &dst = m->t[target][idx];
        dst.resize(nel);
        if (tensor_is_u16(name))
        {
            if (pos + 2 * nel > buf.size())
            {
                delete m;
                return fail("truncated tensor data");
            }
            const unsigned char *q = buf.data() + pos;
            for (size_t i = 0; i < nel; ++i)
            {
                uint16_t v;
                memcpy(&v, q + 2 * i, 2);
                dst[i] = (float)v * scale + offset; /* model.cpp:656-662 */
            }
            pos += 2 * nel;
        }
        else
        {
            if (pos + nel > buf.size())
            {
                delete m;
                return fail("truncated tensor data");
            }
            const unsigned char *q = buf.data() + pos;
            for (size_t i = 0; i < nel; ++i)
                dst[i] = (float)q[i] * scale + offset; /* model.cpp:610-616 */
            pos += nel;
        }
        ++n_loaded;
        if (idx == T_BN3_RV) /* model.cpp:530-539 */
            ++target;
    }
    if (n_loaded != 4 * T_COUNT)
    {
        delete m;
        return fail("model file does not hold 4 x 43 tensors");
    }
    return m;
}

/* ---------------------------------------------------------------- FFT (Eigen::FFT<float>)
 * dsp.cpp:130-139: kissfft backend, flags Speedy|HalfSpectrum|Unscaled.  Restated as a
 * mixed-radix (4,4,4,4,4,2) decimation-in-time complex FFT of 2048 points on the even/odd
 * packed real signal plus the real-FFT split step -- the algorithm kissfft uses for an even
 * real length.  Twiddles are rounded from double; the absent Eigen pin makes bit parity with
 * the reference FFT unobtainable (header: parity unpinned). */
namespace
{
struct fft_plan
{
    int n;
    std::vector<cf> tw;     /* exp(-2 pi i k / n), k < n */
    std::vector<cf> rtw;    /* exp(-2 pi i k / (2n)), k <= n/2: real-FFT split twiddles */
    std::vector<int> radix; /* stage radices */
    explicit fft_plan(int n_) : n(n_), tw(n_), rtw(n_ / 2 + 1)
    {
        for (int k = 0; k < n; ++k)
        {
            double ph = -2.0 * M_PI * (double)k / (double)n;
            tw[k] = cf((float)cos(ph), (float)sin(ph));
        }
        for (int k = 0; k <= n / 2; ++k)
        {
            double ph = -M_PI * (double)k / (double)n;
            rtw[k] = cf((float)cos(ph), (float)sin(ph));
        }
        int m = n;
        while (m % 4 == 0)
        {
            radix.push_back(4);
            m /= 4;
        }
        while (m % 2 == 0)
        {
            radix.push_back(2);
            m /= 2;
        }
    }
    /* out[0..len) = DFT of in[0], in[stride], ...; len = n / stride */
    void work(cf *out, const cf *in, int stride, int stage, bool inverse) const
    {
        int p = radix[stage];
        int len = n / stride; /* length of this sub-transform */
        int m = len / p;
        if (m == 1)
        {
            for (int q = 0; q < p; ++q)
                out[q] = in[q * stride];
        }
        else
        {
            for (int q = 0; q < p; ++q)
                work(out + q * m, in + q * stride, stride * p, stage + 1, inverse);
        }
        auto twd = [&](int idx) -> cf {
            cf w = tw[idx % n];
            return inverse ? std::conj(w) : w;
        };
        if (p == 2)
        {
            for (int k = 0; k < m; ++k)
            {
                cf t = out[k + m] * twd(k * stride);
                cf a = out[k];
                out[k] = a + t;
                out[k + m] = a - t;
            }
        }
        else
        {
            for (int k = 0; k < m; ++k)
            {
                cf a0 = out[k];
                cf a1 = out[k + m] * twd(k * stride);
                cf a2 = out[k + 2 * m] * twd(2 * k * stride);
                cf a3 = out[k + 3 * m] * twd(3 * k * stride);
                cf s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
                /* -i*d13 forward, +i*d13 inverse */
                cf r = inverse ? cf(-d13.imag(), d13.real()) : cf(d13.imag(), -d13.real());
                out[k] = s02 + s13;
                out[k + m] = d02 + r;
                out[k + 2 * m] = s02 - s13;
                out[k + 3 * m] = d02 - r;
            }
        }
    }
};

const fft_plan &plan2048()
{
    static fft_plan p(NFFT / 2);
    return p;
}

void rfft(const float *in, cf *out /*2049*/)
{
    const fft_plan &p = plan2048();
    const int nc = NFFT / 2;
    cf z[NFFT / 2];
    p.work(z, reinterpret_cast<const cf *>(in), 1, 0, false);
    out[0] = cf(z[0].real() + z[0].imag(), 0.0f);
    out[nc] = cf(z[0].real() - z[0].imag(), 0.0f);
    for (int k = 1; k <= nc / 2; ++k)
    {
        cf a = z[k], b = std::conj(z[nc - k]);
        cf e = a + b, o = a - b;
        cf t = o * p.rtw[k]; /* exp(-2 pi i k / 4096) */
        cf to(t.imag(), -t.real()); /* -i * t */
        out[k] = (e + to) * 0.5f;
        out[nc - k] = std::conj(e - to) * 0.5f;
    }
}

void irfft(const cf *in /*2049*/, float *out)
{
    const fft_plan &p = plan2048();
    const int nc = NFFT / 2;
    cf z[NFFT / 2];
    z[0] = cf(in[0].real() + in[nc].real(), in[0].real() - in[nc].real());
    for (int k = 1; k <= nc / 2; ++k)
    {
        cf a = in[k], b = std::conj(in[nc - k]);
        cf e = a + b, o = a - b;
        cf t = o * std::conj(p.rtw[k]);
        cf to(-t.imag(), t.real()); /* +i * t */
        z[k] = e + to;
        z[nc - k] = std::conj(e - to);
    }
    p.work(reinterpret_cast<cf *>(out), z, 1, 0, true);
}
} // namespace

extern "C" void oracle_rfft4096(const float *in, float *out_complex)
{
    rfft(in, reinterpret_cast<cf *>(out_complex));
}
extern "C" void oracle_irfft4096(const float *in_complex, float *out)
{
    irfft(reinterpret_cast<const cf *>(in_complex), out);
}

/* ---------------------------------------------------------------- dsp */
extern "C" int oracle_nb_frames(int n_samples_buf) { return n_samples_buf / HOP + 1; } /* dsp.hpp:48 */

/* dsp.hpp:61-78: periodic Hann with the reference's truncated PI constant */
extern "C" void oracle_hann_window(float *w)
{
    static const float PI = 3.14159265359F;
    float floatN = (float)(NFFT + 1);
    for (int n = 0; n < NFFT; ++n)
        w[n] = 0.5F * (1.0F - cosf(2.0F * PI * (float)n / (floatN - 1)));
}

/* dsp.hpp:80-101 */
extern "C" void oracle_window_sumsq(int nb_frames, float *nw)
{
    std::vector<float> w(NFFT);
    oracle_hann_window(w.data());
    int total = NFFT + HOP * (nb_frames - 1);
    std::fill(nw, nw + total, 0.0f);
    for (int i = 0; i < nb_frames; ++i)
    {
        int s = i * HOP;
        for (int j = s; j < std::min(total, s + NFFT); ++j)
            nw[j] += w[j - s] * w[j - s];
    }
}

/* dsp.cpp:141-176 with pad_signal (109-128) and stft_inner (209-229) */
extern "C" void oracle_stft(const float *audio, int n, int n_buf, float *spec_f)
{
    const int T = oracle_nb_frames(n_buf);
    const int pad = NFFT / 2;
    cf *spec = reinterpret_cast<cf *>(spec_f);
    std::vector<float> w(NFFT);
    oracle_hann_window(w.data());
    /* padded_waveform_mono_in is value-initialised once (dsp.hpp:52) and, because the struct
     * is passed by value per segment (inference.hpp:22), starts from zeros every call; the
     * same buffer is then reused for channel 1 after channel 0. */
    std::vector<float> padded((size_t)n_buf + NFFT, 0.0f);
    for (int ch = 0; ch < 2; ++ch)
    {
        for (int i = 0; i < n; ++i) /* dsp.cpp:154-155 */
            padded[pad + i] = audio[2 * (size_t)i + ch];
        /* pad_signal, dsp.cpp:109-128: mirror INCLUDING the edge sample (numpy 'symmetric') */
        {
            size_t sz = padded.size();
            std::vector<float> head(padded.begin() + pad, padded.begin() + 2 * pad);
            std::vector<float> tail(padded.begin() + (sz - 2 * pad), padded.begin() + (sz - pad));
            for (int i = 0; i < pad; ++i)
            {
                padded[i] = head[pad - 1 - i];
                padded[sz - pad + i] = tail[pad - 1 - i];
            }
        }
#pragma omp parallel for num_threads(nthreads()) schedule(static)
        for (int f = 0; f < T; ++f) /* dsp.cpp:214-228 */
        {
            float frame[NFFT];
            cf bins[NB];
            const float *src = padded.data() + (size_t)f * HOP;
            for (int i = 0; i < NFFT; ++i)
                frame[i] = src[i] * w[i];
            rfft(frame, bins);
            for (int b = 0; b < NB; ++b) /* dsp.cpp:168-174 scatter into spec(c,f,b) */
                spec[ch + 2 * ((size_t)f + (size_t)T * b)] = bins[b];
        }
    }
}

/* dsp.cpp:178-207 and istft_inner 231-258 */
extern "C" void oracle_istft(const float *spec_f, int n, int n_buf, float *out)
{
    const int T = oracle_nb_frames(n_buf);
    const int pad = NFFT / 2;
    const cf *spec = reinterpret_cast<const cf *>(spec_f);
    std::vector<float> w(NFFT);
    oracle_hann_window(w.data());
    std::vector<float> nw((size_t)NFFT + (size_t)HOP * (T - 1));
    oracle_window_sumsq(T, nw.data());
    std::vector<float> frames((size_t)T * NFFT);
    std::vector<float> padded((size_t)n_buf + NFFT);
    for (int ch = 0; ch < 2; ++ch)
    {
#pragma omp parallel for num_threads(nthreads()) schedule(static)
        for (int f = 0; f < T; ++f)
        {
            cf bins[NB];
            for (int b = 0; b < NB; ++b)
                bins[b] = spec[ch + 2 * ((size_t)f + (size_t)T * b)];
            irfft(bins, frames.data() + (size_t)f * NFFT); /* cfg.inv, Unscaled */
        }
        std::fill(padded.begin(), padded.end(), 0.0f); /* dsp.cpp:234-235 */
        for (int f = 0; f < T; ++f) /* frame order matters for the fp32 overlap-add sum */
        {
            size_t start = (size_t)f * HOP;
            const float *fr = frames.data() + (size_t)f * NFFT;
            for (int i = 0; i < NFFT; ++i) /* dsp.cpp:248-256, same operation order */
                padded[start + i] +=
                    fr[i] * w[i] * 1.0f / float(NFFT) / (nw[start + i] + 1e-8f);
        }
        for (int i = 0; i < n; ++i) /* dsp.cpp:203-205: row assign keeps n columns (NDEBUG) */
            out[2 * (size_t)i + ch] = padded[pad + i];
    }
}

/* ---------------------------------------------------------------- dense helpers */
namespace
{
/* fp32 dot product with 16 independent partial sums (vectorisable without -ffast-math).
 * The reference routes these products to BLAS sgemm/sgemv whose summation order is
 * unspecified (parity unpinned); this fixes one definite order for the oracle. */
inline float dotf(const float *a, const float *b, int n)
{
    float acc[16] = {0};
    int i = 0;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; ++j)
            acc[j] += a[i + j] * b[i + j];
    float tail = 0.0f;
    for (; i < n; ++i)
        tail += a[i] * b[i];
    for (int s = 8; s >= 1; s >>= 1)
        for (int j = 0; j < s; ++j)
            acc[j] += acc[j + s];
    return acc[0] + tail;
}

/* Y (T x N) = X (T x K) * W^T, W in PyTorch layout (N x K) row-major == the reference's
 * ColMajor (K x N) matrix (model.cpp:578-619 reads the raw bytes straight into it). */
void matmul_wt(const float *X, const float *W, float *Y, int T, int K, int N)
{
#pragma omp parallel for num_threads(nthreads()) schedule(static)
    for (int t = 0; t < T; ++t)
        for (int o = 0; o < N; ++o)
            Y[(size_t)t * N + o] = dotf(X + (size_t)t * K, W + (size_t)o * K, K);
}

inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); } /* lstm.cpp:36-39 */
} // namespace

extern "C" size_t oracle_stream_state_floats(int hidden) { return (size_t)4 * 3 * 2 * 2 * (hidden / 2); }

/* lstm.cpp:101-179.  Per-timestep GEMVs exactly like the reference (the input projection is
 * NOT batched), gate order i|f|g|o (lstm.cpp:143-152), expression order
 * ((W_ih x + b_ih) + W_hh h) + b_hh (lstm.cpp:132-140). */
extern "C" void oracle_lstm_forward(const oracle_model *m, int target, const float *input, int T,
                                    float *state, float *out)
{
    const int H = m->hidden, Hl = H / 2, G = 4 * Hl;
    std::vector<float> cur(input, input + (size_t)T * H), nxt((size_t)T * H);
    for (int layer = 0; layer < 3; ++layer)
    {
#pragma omp parallel for num_threads(std::min(2, nthreads())) schedule(static)
        for (int dir = 0; dir < 2; ++dir)
        {
            const float *Wih = m->lstm(target, layer, dir, 0), *Whh = m->lstm(target, layer, dir, 1);
            const float *bih = m->lstm(target, layer, dir, 2), *bhh = m->lstm(target, layer, dir, 3);
            float *h = state + ((size_t)(layer * 2 + dir) * 2 + 0) * Hl;
            float *c = state + ((size_t)(layer * 2 + dir) * 2 + 1) * Hl;
            std::vector<float> gates(G), hn(Hl);
            for (int step = 0; step < T; ++step)
            {
                int t = dir == 0 ? step : T - 1 - step; /* lstm.cpp:118-120 */
                const float *x = cur.data() + (size_t)t * H;
                for (int o = 0; o < G; ++o)
                    gates[o] = ((dotf(Wih + (size_t)o * H, x, H) + bih[o]) +
                                dotf(Whh + (size_t)o * Hl, h, Hl)) +
                               bhh[o];
                for (int j = 0; j < Hl; ++j)
                {
                    float i_t = sigmoidf(gates[j]);
                    float f_t = sigmoidf(gates[Hl + j]);
                    float g_t = tanhf(gates[2 * Hl + j]);
                    float o_t = sigmoidf(gates[3 * Hl + j]);
                    float c_t = f_t * c[j] + i_t * g_t; /* lstm.cpp:154-156 */
                    c[j] = c_t;
                    hn[j] = o_t * tanhf(c_t); /* lstm.cpp:157 */
                }
                memcpy(h, hn.data(), sizeof(float) * Hl);
                /* lstm.cpp:163-164 + 170-171: fwd in columns [0,Hl), bwd in [Hl,2Hl) */
                memcpy(nxt.data() + (size_t)t * H + (size_t)dir * Hl, hn.data(), sizeof(float) * Hl);
            }
        }
        cur.swap(nxt);
    }
    memcpy(out, cur.data(), sizeof(float) * (size_t)T * H);
}

/* inference.cpp:70-186, one target */
extern "C" void oracle_target_network(const oracle_model *m, int target, const float *x,
                                      const float *mix_mag, int T, float *state, float *fc1_out,
                                      float *lstm_out, float *mask, float *target_mag)
{
    oracle_target_network_ex(m, target, x, mix_mag, T, state, fc1_out, lstm_out, nullptr, mask, target_mag);
}

extern "C" void oracle_target_network_ex(const oracle_model *m, int target, const float *x,
                                         const float *mix_mag, int T, float *state, float *fc1_out,
                                         float *lstm_out, float *fc2_out, float *mask, float *target_mag)
{
    const int H = m->hidden;
    /* inference.cpp:75-83 with the duplicated mean/scale of model.cpp:240-264 (F8 order) */
    std::vector<float> xin((size_t)T * NIN);
    const float *im = m->p(target, T_INPUT_MEAN), *is = m->p(target, T_INPUT_SCALE);
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < NIN; ++j)
            xin[(size_t)t * NIN + j] = x[(size_t)t * NIN + j] * is[j % CROP] + im[j % CROP];
    /* fc1 (no bias) inference.cpp:86, bn1 + tanh inference.cpp:91-99 */
    std::vector<float> a1((size_t)T * H);
    matmul_wt(xin.data(), m->p(target, T_FC1_W), a1.data(), T, NIN, H);
    {
        const float *w = m->p(target, T_BN1_W), *b = m->p(target, T_BN1_B);
        const float *rm = m->p(target, T_BN1_RM), *rv = m->p(target, T_BN1_RV);
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < H; ++j)
            {
                float y = a1[(size_t)t * H + j];
                a1[(size_t)t * H + j] = tanhf(((y - rm[j]) / sqrtf(rv[j] + 1e-5f)) * w[j] + b[j]);
            }
    }
    if (fc1_out)
        memcpy(fc1_out, a1.data(), sizeof(float) * a1.size());
    /* lstm, inference.cpp:108-110 */
    std::vector<float> lo((size_t)T * H);
    oracle_lstm_forward(m, target, a1.data(), T, state, lo.data());
    if (lstm_out)
        memcpy(lstm_out, lo.data(), sizeof(float) * lo.size());
    /* skip concat inference.cpp:118-123; fc2 :127; bn2+relu :132-140 */
    std::vector<float> cat((size_t)T * 2 * H);
    for (int t = 0; t < T; ++t)
    {
        memcpy(&cat[(size_t)t * 2 * H], &a1[(size_t)t * H], sizeof(float) * H);
        memcpy(&cat[(size_t)t * 2 * H + H], &lo[(size_t)t * H], sizeof(float) * H);
    }
    std::vector<float> a2((size_t)T * H);
    matmul_wt(cat.data(), m->p(target, T_FC2_W), a2.data(), T, 2 * H, H);
    {
        const float *w = m->p(target, T_BN2_W), *b = m->p(target, T_BN2_B);
        const float *rm = m->p(target, T_BN2_RM), *rv = m->p(target, T_BN2_RV);
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < H; ++j)
            {
                float y = a2[(size_t)t * H + j];
                a2[(size_t)t * H + j] =
                    std::max(0.0f, ((y - rm[j]) / sqrtf(rv[j] + 1e-5f)) * w[j] + b[j]);
            }
    }
    if (fc2_out)
        memcpy(fc2_out, a2.data(), sizeof(float) * a2.size());
    /* fc3 :143; bn3 :148-156 (no activation); output scale + relu :161-166 */
    std::vector<float> a3((size_t)T * NOUT);
    matmul_wt(a2.data(), m->p(target, T_FC3_W), a3.data(), T, H, NOUT);
    {
        const float *w = m->p(target, T_BN3_W), *b = m->p(target, T_BN3_B);
        const float *rm = m->p(target, T_BN3_RM), *rv = m->p(target, T_BN3_RV);
        const float *om = m->p(target, T_OUTPUT_MEAN), *os = m->p(target, T_OUTPUT_SCALE);
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < NOUT; ++j)
            {
                float y = a3[(size_t)t * NOUT + j];
                y = ((y - rm[j]) / sqrtf(rv[j] + 1e-5f)) * w[j] + b[j];
                a3[(size_t)t * NOUT + j] = std::max(0.0f, y * os[j % NB] + om[j % NB]);
            }
    }
    if (mask)
        memcpy(mask, a3.data(), sizeof(float) * a3.size());
    /* mask * mix_mag, inference.cpp:173-183 */
    if (target_mag)
        for (int t = 0; t < T; ++t)
            for (int b = 0; b < NB; ++b)
                for (int ch = 0; ch < 2; ++ch)
                {
                    size_t k = ch + 2 * ((size_t)t + (size_t)T * b);
                    target_mag[k] = a3[(size_t)t * NOUT + b + (size_t)NB * ch] * mix_mag[k];
                }
}

/* ---------------------------------------------------------------- Wiener (wiener.cpp:92-425)
 * Arithmetic order follows the reference loop by loop, including F5 (Re+Im squared, :187-202)
 * and F6 (sqrt(eps) added once per source, :311-323); the 13 MB temporaries are not
 * materialised but every value is formed by the same sequence of fp32 operations. */
extern "C" void oracle_wiener(float *mix_f, const float *const *target_mags, int T,
                              float *const *y_out)
{
    const float EPS = 1e-10f, SCALE = 10.0f; /* wiener.hpp:12-13 */
    const int BATCH = 200;                  /* wiener.hpp:16 */
    cf *X = reinterpret_cast<cf *>(mix_f);
    const size_t NEL = (size_t)2 * T * NB;
    auto I3 = [&](int c, int f, int b) -> size_t { return c + 2 * ((size_t)f + (size_t)T * b); };
    cf *y[4];
    for (int s = 0; s < 4; ++s)
        y[s] = reinterpret_cast<cf *>(y_out[s]);
    /* :96-109 phase = arg(X); y = polar(mag, phase) (dsp.cpp:260-289) */
    for (int s = 0; s < 4; ++s)
#pragma omp parallel for num_threads(nthreads()) schedule(static)
        for (size_t k = 0; k < NEL; ++k)
        {
            float ph = std::arg(X[k]);
            y[s][k] = std::polar(target_mags[s][k], ph);
        }
    /* find_max_abs :37-52 */
    float mx = -1.0f;
    for (size_t k = 0; k < NEL; ++k)
        mx = std::max(mx, std::sqrt(std::norm(X[k])));
    const float max_abs = std::max(1.0f, mx / SCALE);
    /* :118-146 */
    for (size_t k = 0; k < NEL; ++k)
        X[k] = cf(X[k].real() / max_abs, X[k].imag() / max_abs);
    for (int s = 0; s < 4; ++s)
        for (size_t k = 0; k < NEL; ++k)
            y[s][k] = cf(y[s][k].real() / max_abs, y[s][k].imag() / max_abs);

    const float reg = std::sqrt(EPS); /* :161-165 */
    std::vector<float> v((size_t)T * NB * 4);
    auto V = [&](int f, int b, int s) -> float & { return v[((size_t)f * NB + b) * 4 + s]; };
    /* :181-205 (F5) */
#pragma omp parallel for num_threads(nthreads()) schedule(static)
    for (int f = 0; f < T; ++f)
        for (int b = 0; b < NB; ++b)
            for (int s = 0; s < 4; ++s)
            {
                float sum = 0.0f;
                for (int c = 0; c < 2; ++c)
                {
                    float re = 0.0f, im = 0.0f;
                    re += y[s][I3(c, f, b)].real();
                    re += y[s][I3(c, f, b)].imag();
                    sum += (re * re) + (im * im);
                }
                V(f, b, s) = sum / 2;
            }
    /* :207-274 spatial covariance R[s](b, c1, c2) and weight */
    std::vector<cf> R((size_t)4 * NB * 4);
    auto Rr = [&](int s, int b, int c1, int c2) -> cf & { return R[(((size_t)s * NB + b) * 2 + c1) * 2 + c2]; };
    for (int s = 0; s < 4; ++s)
    {
#pragma omp parallel for num_threads(nthreads()) schedule(static)
        for (int b = 0; b < NB; ++b)
        {
            cf Racc[2][2] = {{0, 0}, {0, 0}};
            float weight = EPS;
            for (int pos = 0; pos < T; pos += BATCH)
            {
                int t_end = std::min(T, pos + BATCH);
                cf tmp[2][2] = {{0, 0}, {0, 0}};
                for (int f = pos; f < t_end; ++f)
                    for (int c1 = 0; c1 < 2; ++c1)
                        for (int c2 = 0; c2 < 2; ++c2)
                        {
                            /* calculateCovariance :435-478: Cj = 0 + a*conj(b) */
                            cf a = y[s][I3(c1, f, b)], bb = std::conj(y[s][I3(c2, f, b)]);
                            cf prod(a.real() * bb.real() - a.imag() * bb.imag(),
                                    a.real() * bb.imag() + a.imag() * bb.real());
                            cf cj = cf(0, 0) + prod;
                            tmp[c1][c2] += cj; /* :228-240 */
                        }
                for (int c1 = 0; c1 < 2; ++c1)
                    for (int c2 = 0; c2 < 2; ++c2)
                        Racc[c1][c2] += tmp[c1][c2]; /* :243 */
                for (int f = pos; f < t_end; ++f)
                    weight += V(f, b, s); /* :247-253 */
            }
            for (int c1 = 0; c1 < 2; ++c1)
                for (int c2 = 0; c2 < 2; ++c2) /* :259-269 complex / float */
                    Rr(s, b, c1, c2) = cf(Racc[c1][c2].real() / weight, Racc[c1][c2].imag() / weight);
        }
    }
    auto cmul = [](cf a, cf b) -> cf {
        return cf(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real());
    };
    /* :276-404 */
#pragma omp parallel for num_threads(nthreads()) schedule(static)
    for (int f = 0; f < T; ++f)
        for (int b = 0; b < NB; ++b)
        {
            cf C[2][2] = {{0, 0}, {0, 0}};
            for (int s = 0; s < 4; ++s) /* :307-325 (F6) */
            {
                float mult = V(f, b, s);
                for (int c1 = 0; c1 < 2; ++c1)
                    for (int c2 = 0; c2 < 2; ++c2)
                    {
                        cf r = Rr(s, b, c1, c2);
                        cf term = cf(c1 == c2 ? reg : 0.0f, 0.0f) + cf(mult * r.real(), mult * r.imag());
                        C[c1][c2] += term;
                    }
            }
            /* invert4D :54-84 */
            cf a = C[0][0], bq = C[0][1], c = C[1][0], d = C[1][1];
            cf det = cmul(a, d) - cmul(bq, c);
            float nrm = det.real() * det.real() + det.imag() * det.imag();
            cf invDet(det.real() / nrm, -det.imag() / nrm);
            cf Ci[2][2];
            Ci[0][0] = cmul(invDet, d);
            Ci[0][1] = cmul(-invDet, bq);
            Ci[1][0] = cmul(-invDet, c);
            Ci[1][1] = cmul(invDet, a);
            cf x0 = X[I3(0, f, b)], x1 = X[I3(1, f, b)];
            for (int s = 0; s < 4; ++s)
            {
                cf g[2][2];
                for (int c1 = 0; c1 < 2; ++c1) /* :343-361 */
                    for (int c2 = 0; c2 < 2; ++c2)
                    {
                        cf acc(0, 0);
                        for (int c3 = 0; c3 < 2; ++c3)
                            acc += cmul(Rr(s, b, c1, c3), Ci[c3][c2]);
                        float vv = V(f, b, s); /* :364-376 */
                        g[c1][c2] = cf(acc.real() * vv, acc.imag() * vv);
                    }
                cf out[2] = {cf(0, 0), cf(0, 0)}; /* y zeroed :287-297 */
                for (int c1 = 0; c1 < 2; ++c1)   /* :381-400 */
                    for (int c2 = 0; c2 < 2; ++c2)
                        out[c2] = out[c2] + cmul(g[c2][c1], c1 == 0 ? x0 : x1);
                y[s][I3(0, f, b)] = out[0];
                y[s][I3(1, f, b)] = out[1];
            }
        }
    /* :408-422 */
    for (int s = 0; s < 4; ++s)
        for (size_t k = 0; k < NEL; ++k)
            y[s][k] = cf(y[s][k].real() * max_abs, y[s][k].imag() * max_abs);
}

/* ---------------------------------------------------------------- umx_inference */
extern "C" void oracle_umx_inference(const oracle_model *m, const float *audio, int n, int n_buf,
                                     float *state, float *const *out, int flags, oracle_taps *taps)
{
    const int T = oracle_nb_frames(n_buf), H = m->hidden, Hl = H / 2;
    const size_t NEL = (size_t)2 * T * NB;
    std::vector<float> spec(NEL * 2);
    oracle_stft(audio, n, n_buf, spec.data()); /* inference.cpp:26 */
    const cf *S = reinterpret_cast<const cf *>(spec.data());
    std::vector<float> mix_mag(NEL);
    for (size_t k = 0; k < NEL; ++k)
        mix_mag[k] = std::abs(S[k]); /* inference.cpp:29 */
    std::vector<float> x((size_t)T * NIN);
    for (int t = 0; t < T; ++t) /* inference.cpp:58-68 */
        for (int j = 0; j < CROP; ++j)
        {
            x[(size_t)t * NIN + j] = mix_mag[0 + 2 * ((size_t)t + (size_t)T * j)];
            x[(size_t)t * NIN + j + CROP] = mix_mag[1 + 2 * ((size_t)t + (size_t)T * j)];
        }
    if (taps)
    {
        if (taps->spec)
            memcpy(taps->spec, spec.data(), sizeof(float) * spec.size());
        if (taps->mix_mag)
            memcpy(taps->mix_mag, mix_mag.data(), sizeof(float) * NEL);
        if (taps->x)
            memcpy(taps->x, x.data(), sizeof(float) * x.size());
    }
    std::vector<std::vector<float>> tm(4, std::vector<float>(NEL, 0.0f));
    const int skip = (flags >> 8) & 0xF;
    const int outer = std::min(4, nthreads_total());
    const int inner = std::max(1, nthreads_total() / outer);
#ifdef _OPENMP
    omp_set_max_active_levels(3);
#endif
#pragma omp parallel for num_threads(outer) schedule(static, 1)
    for (int tg = 0; tg < 4; ++tg) /* inference.cpp:70; targets are independent */
    {
        if (skip & (1 << tg))
            continue;
        tl_inner = inner;
        oracle_target_network_ex(m, tg, x.data(), mix_mag.data(), T, state + (size_t)tg * 12 * Hl,
                                 taps ? taps->fc1_out[tg] : nullptr, taps ? taps->lstm_out[tg] : nullptr,
                                 taps ? taps->fc2_out[tg] : nullptr, taps ? taps->mask[tg] : nullptr, tm[tg].data());
    }
    if (taps)
        for (int tg = 0; tg < 4; ++tg)
            if (taps->target_mag[tg])
                memcpy(taps->target_mag[tg], tm[tg].data(), sizeof(float) * NEL);
    std::vector<std::vector<float>> y(4, std::vector<float>(NEL * 2));
    float *yp[4] = {y[0].data(), y[1].data(), y[2].data(), y[3].data()};
    if (flags & 1)
    {
        /* config 2 of BASELINE.json ("no Wiener"): mix-phase estimate only, wiener.cpp:96-109 */
        for (int tg = 0; tg < 4; ++tg)
        {
            cf *yy = reinterpret_cast<cf *>(yp[tg]);
            for (size_t k = 0; k < NEL; ++k)
                yy[k] = std::polar(tm[tg][k], std::arg(S[k]));
        }
    }
    else
    {
        const float *tmp[4] = {tm[0].data(), tm[1].data(), tm[2].data(), tm[3].data()};
        oracle_wiener(spec.data(), tmp, T, yp); /* inference.cpp:192-193 */
    }
    if (taps)
        for (int tg = 0; tg < 4; ++tg)
            if (taps->y[tg])
                memcpy(taps->y[tg], yp[tg], sizeof(float) * NEL * 2);
    for (int tg = 0; tg < 4; ++tg) /* inference.cpp:199-204 */
        oracle_istft(yp[tg], n, n_buf, out[tg]);
}

/* ---------------------------------------------------------------- split / shift drivers */
extern "C" void oracle_split_inference(const oracle_model *m, const float *audio, int length,
                                       int segment_samples, float *const *out, int flags)
{
    const int N = segment_samples;
    const int stride = (int)((1 - 0.25f) * N); /* umx.cpp:181, OVERLAP inference.hpp:15 */
    std::vector<float> state(oracle_stream_state_floats(m->hidden), 0.0f); /* umx.cpp:167-171 */
    std::vector<float> weight(N), sum_w((size_t)length, 0.0f);             /* F4: zeroed fully */
    for (int i = 0; i < N / 2; ++i) /* umx.cpp:199-203 */
    {
        weight[i] = (float)(i + 1);
        weight[N - i - 1] = (float)(i + 1);
    }
    float wmax = *std::max_element(weight.begin(), weight.end());
    for (int i = 0; i < N; ++i)
        weight[i] = std::pow(weight[i] / wmax, 1.0f); /* umx.cpp:205-206 */
    for (int tg = 0; tg < 4; ++tg)
        std::fill(out[tg], out[tg] + (size_t)2 * length, 0.0f);
    std::vector<std::vector<float>> chunk_out(4);
    for (int offset = 0; offset < length; offset += stride) /* umx.cpp:214 */
    {
        int chunk_len = std::min(N, length - offset);
        for (int tg = 0; tg < 4; ++tg)
            chunk_out[tg].assign((size_t)2 * chunk_len, 0.0f);
        float *co[4] = {chunk_out[0].data(), chunk_out[1].data(), chunk_out[2].data(),
                        chunk_out[3].data()};
        oracle_umx_inference(m, audio + (size_t)2 * offset, chunk_len, N, state.data(), co, flags,
                             nullptr);
        for (int tg = 0; tg < 4; ++tg) /* umx.cpp:234-249 */
            for (int k = 0; k < N && offset + k < length; ++k)
                for (int c = 0; c < 2; ++c)
                    out[tg][2 * (size_t)(offset + k) + c] +=
                        weight[k % chunk_len] * chunk_out[tg][2 * (size_t)k + c];
        for (int k = 0; k < N && offset + k < length; ++k) /* umx.cpp:253-260 */
            sum_w[offset + k] += weight[k % chunk_len];
    }
    for (int tg = 0; tg < 4; ++tg) /* umx.cpp:264-273 */
        for (int k = 0; k < length; ++k)
            for (int c = 0; c < 2; ++c)
                out[tg][2 * (size_t)k + c] /= sum_w[k];
}

extern "C" void oracle_shift_inference(const oracle_model *m, const float *audio, int length,
                                       int segment_samples, int offset, float *const *out, int flags)
{
    const int max_shift = (int)(0.5f * 44100); /* umx.cpp:112-113 */
    /* umx.cpp:120-122: length + max_shift - offset; the reference then writes [offset, offset + length), out of
     * bounds for offset > max_shift / 2 (never reached by its unseeded rand(): 4033) -- sized to hold the block */
    const int L2 = length + std::max(max_shift - offset, offset);
    std::vector<float> shifted((size_t)2 * L2, 0.0f);
    memcpy(shifted.data() + (size_t)2 * offset, audio, sizeof(float) * 2 * (size_t)length);
    std::vector<std::vector<float>> o(4, std::vector<float>((size_t)2 * L2));
    float *op[4] = {o[0].data(), o[1].data(), o[2].data(), o[3].data()};
    oracle_split_inference(m, shifted.data(), L2, segment_samples, op, flags);
    for (int tg = 0; tg < 4; ++tg) /* umx.cpp:136-147 */
        memcpy(out[tg], o[tg].data() + (size_t)2 * offset, sizeof(float) * 2 * (size_t)length);
}

extern "C" int oracle_num_threads(void) { return nthreads_total(); }
extern "C" void oracle_set_num_threads(int n) { g_threads = n; }
