// stft_kernels.h -- STFT / iSTFT kernels (replace dsp.cpp:109-258 + inference.cpp:29,52-68).
//
// HBM layouts (b fastest, so a wave touches consecutive bins):
//   spec    float2 [2][T][2049]      (reference: Eigen ColMajor (2,T,2049), c fastest)
//   (|X| is not stored: x holds the cropped part the network reads, the Wiener kernels form the rest from spec)
//   x       float  [Tp][KX]          row t = [ |L|[0:1487] | |R|[0:1487] | 0 0 ]  (inference.cpp:58-68)
//   y       float2 [4][2][T][2049]   per-source complex spectrograms
//   frames  float2 [4][T][4096]      (.x = left, .y = right) windowed + normalised iFFT frames
// Algorithmic HBM bytes per 60 s segment (T = 2584):
//   stft : read 21.2 MB audio; write 84.7 (spec) + 30.8 (x) = 115.5 MB
//   istft: read 338.8 MB (y), write 338.7 MB (frames); ola: read 338.7 MB, write 84.7 MB
#pragma once
#include "fft4096.h"

namespace umx
{

// One workgroup per STFT_RUN consecutive frames; both channels in one complex FFT.  grid (ceil(T / STFT_RUN), lanes): entry blockIdx.y of
// `in`; spec, x and maxabs_bits are lane 0's, lane l's sit l strides behind.
// What bounds the kernel is the CU's one vector-memory address path (~24 cycles per wave instruction whatever its width,
// tools/vmem_issue_probe.hip): a frame per workgroup issued 80 per thread -- 16 samples, 16 window values, 30 twiddles, 18 stores.  The
// window rides in registers over the run's frames, the second pass's twiddles come from LDS and a thread stores pairs of adjacent bins
// (round 6): 48 + 32 / STFT_RUN per frame.
#ifndef STFT_RUN_FRAMES
#define STFT_RUN_FRAMES 4
#endif
constexpr int STFT_RUN = STFT_RUN_FRAMES;
struct StftIn
{
    const float *audio[MAX_TRACK_LANES]; // per entry of lanes
    int n[MAX_TRACK_LANES];
    LaneSet lanes;
};
__global__ __launch_bounds__(256) void stft_kernel(StftIn in, int N, int T,
                                                   const float *__restrict__ window,
                                                   const float2 *__restrict__ tw1,
                                                   const float2 *__restrict__ tw2,
                                                   float2 *__restrict__ spec, size_t spec_stride, float *__restrict__ x, size_t x_stride,
                                                   unsigned *__restrict__ maxabs_bits)
{
    __shared__ float2 buf[FFT_LDS_ELEMS];
    __shared__ float2 tw1s[256];
    __shared__ float red[4];
    const int j = threadIdx.x;
    const int f0 = (int)blockIdx.x * STFT_RUN, f1 = min(T, f0 + STFT_RUN);
    const int ln = in.lanes.id[blockIdx.y], n = in.n[blockIdx.y];
    const float *__restrict__ audio = in.audio[blockIdx.y];
    spec += (size_t)ln * spec_stride;
    x += (size_t)ln * x_stride;
    maxabs_bits += ln;
    const float2 *a2 = reinterpret_cast<const float2 *>(audio);
    tw1s[j] = tw1[j]; // (read behind the first barrier of the first transform)
    const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(spec, 0, (int)((size_t)2 * T * NBINS * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(x, 0, (int)((size_t)T * KX * 4), 0x00020000);
    float wv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        wv[r] = window[j + 256 * r];
    float lmax = 0.f;
    for (int f = f0; f < f1; ++f)
    {
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const int i = j + 256 * r;
            const int p = f * HOP + i; // index into the reference's padded buffer (size N + 4096)
            int d;                     // index into the chunk; pad_signal dsp.cpp:109-128 (symmetric)
            if (p < NFFT / 2)
                d = NFFT / 2 - 1 - p;
            else if (p < N + NFFT / 2)
                d = p - NFFT / 2;
            else
                d = N - 1 - (p - (N + NFFT / 2));
            float2 s = (d < n) ? a2[d] : make_float2(0.f, 0.f); // short chunk: rest of buffer is zeros
            const float w = wv[r];
            v[r] = make_float2(s.x * w, s.y * w); // dsp.cpp:220-225
        }
        fft4096<false>(v, buf, tw1s, tw2);
        // a thread takes PAIRS of adjacent bins (2 j + 512 i, + 1): one 16-byte store per channel for the pair's spectrogram and one 8-byte
        // store for its |X| instead of two each -- 16 + 1 instead of 27 store instructions per thread and frame (buffer stores: dword alignment
        // is all they need -- a row of 2049 complex bins starts on an odd 8 bytes every other frame)
        auto one_bin = [&](int k, float2 &sL, float2 &sR) {
            const float2 zk = buf[fft_pad(k)];
            const float2 zn = cconj(buf[fft_pad((NFFT - k) & (NFFT - 1))]);
            sL = make_float2((zk.x + zn.x) * 0.5f, (zk.y + zn.y) * 0.5f);
            const float2 dd = csub(zk, zn);
            sR = make_float2(dd.y * 0.5f, -dd.x * 0.5f); // (zk - zn) / (2i)
            // wiener.cpp:37-52 find_max_abs uses sqrt(norm(z))
            lmax = fmaxf(lmax, fmaxf(sqrtf(sL.x * sL.x + sL.y * sL.y), sqrtf(sR.x * sR.x + sR.y * sR.y)));
        };
        // |X| (inference.cpp:29 abs()) is kept only where the network reads it (k < CROP); the Wiener kernels that need it for every bin
        // have the spectrogram in registers anyway and form it again (mix_magnitude, common.h)
        const int rowL = ((0 * T + f) * NBINS) * 8, rowR = ((1 * T + f) * NBINS) * 8; // bytes
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int k = 2 * j + 512 * i;
            float2 sL0, sR0, sL1, sR1;
            one_bin(k, sL0, sR0);
            one_bin(k + 1, sL1, sR1);
            typedef unsigned st_u4 __attribute__((ext_vector_type(4)));
            typedef unsigned st_u2 __attribute__((ext_vector_type(2)));
            const st_u4 pl = {__float_as_uint(sL0.x), __float_as_uint(sL0.y), __float_as_uint(sL1.x), __float_as_uint(sL1.y)};
            const st_u4 pr = {__float_as_uint(sR0.x), __float_as_uint(sR0.y), __float_as_uint(sR1.x), __float_as_uint(sR1.y)};
            __builtin_amdgcn_raw_buffer_store_b128(pl, rs_spec, k * 8, rowL, 2); // (nt: read next by the Wiener kernels, a whole network later)
            __builtin_amdgcn_raw_buffer_store_b128(pr, rs_spec, k * 8, rowR, 2);
            const float aL0 = mix_magnitude(sL0), aL1 = mix_magnitude(sL1), aR0 = mix_magnitude(sR0), aR1 = mix_magnitude(sR1);
            if (k + 1 < CROP)
            {
                const st_u2 xl = {__float_as_uint(aL0), __float_as_uint(aL1)}, xr = {__float_as_uint(aR0), __float_as_uint(aR1)};
                __builtin_amdgcn_raw_buffer_store_b64(xl, rs_x, k * 4, f * KX * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b64(xr, rs_x, k * 4, (f * KX + CROP) * 4, 0);
            }
            else if (k < CROP) // (CROP is odd: the last kept bin has no partner)
            {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(aL0), rs_x, k * 4, f * KX * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(aR0), rs_x, k * 4, (f * KX + CROP) * 4, 0);
            }
        }
        if (j == 0) // the Nyquist bin
        {
            float2 sL, sR;
            one_bin(NFFT / 2, sL, sR);
            stream_store2(spec + ((size_t)0 * T + f) * NBINS + NFFT / 2, sL);
            stream_store2(spec + ((size_t)1 * T + f) * NBINS + NFFT / 2, sR);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the transform's buffer is free for the next frame (LDS only: the stores need no wait)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((j & 63) == 0)
        red[j >> 6] = lmax;
    __syncthreads();
    if (j == 0)
    {
        float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(maxabs_bits, __float_as_uint(m)); // non-negative floats order like uints
    }
}

// grid (T, 4 sources).  Unscaled inverse real FFT of both channels at once, then the reference's
// per-sample weight  frame*w * 1.0f / 4096 / (nw + 1e-8f)  (dsp.cpp:248-256, same op order).
__global__ __launch_bounds__(256) void istft_frames_kernel(const float2 *__restrict__ y, int T,
                                                           const float *__restrict__ window,
                                                           const float *__restrict__ nw,
                                                           const float2 *__restrict__ tw1,
                                                           const float2 *__restrict__ tw2,
                                                           float2 *__restrict__ frames)
{
    __shared__ float2 buf[FFT_LDS_ELEMS];
    const int f = blockIdx.x, src = blockIdx.y, j = threadIdx.x;
    const float2 *yL = y + (((size_t)src * 2 + 0) * T + f) * NBINS;
    const float2 *yR = y + (((size_t)src * 2 + 1) * T + f) * NBINS;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        const int i = j + 256 * r;
        if (i <= NFFT / 2)
        {
            float2 a = yL[i], b = yR[i];
            if (i == 0 || i == NFFT / 2) // a real inverse FFT ignores Im of DC / Nyquist
            {
                a.y = 0.f;
                b.y = 0.f;
            }
            v[r] = make_float2(a.x - b.y, a.y + b.x); // a + i b
        }
        else
        {
            const float2 a = yL[NFFT - i], b = yR[NFFT - i];
            v[r] = make_float2(a.x + b.y, b.x - a.y); // conj(a) + i conj(b)
        }
    }
    fft4096<true>(v, buf, tw1, tw2);
    float2 *dst = frames + ((size_t)src * T + f) * NFFT;
    const size_t start = (size_t)f * HOP;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        const int i = j + 256 * r;
        const float2 z = buf[fft_pad(i)];
        const float w = window[i];
        const float den = nw[start + i] + 1e-8f;
        dst[i] = make_float2(z.x * w * 1.0f / float(NFFT) / den, z.y * w * 1.0f / float(NFFT) / den);
    }
}

// Overlap-add in ascending frame order (the reference's fp32 summation order, dsp.cpp:237-257)
// and crop [2048, 2048+n) (dsp.cpp:203-205).  out: 4 x (2,n) interleaved.  grid (ceil(n/256), 4).
// grid (ceil(max n / 256), 4, lanes): entry blockIdx.z of `out`; frames is lane 0's.
struct OlaOut
{
    float *p[MAX_TRACK_LANES][4]; // per entry of lanes
    int n[MAX_TRACK_LANES];
    LaneSet lanes;
};
__global__ __launch_bounds__(256) void istft_ola_kernel(const float2 *__restrict__ frames, size_t frames_stride, int T, OlaOut out)
{
    const int s = blockIdx.x * 256 + threadIdx.x, src = blockIdx.y, n = out.n[blockIdx.z];
    if (s >= n)
        return;
    frames += (size_t)out.lanes.id[blockIdx.z] * frames_stride;
    const int p = s + NFFT / 2;
    const int f_hi = min(T - 1, p / HOP);
    const int f_lo = p >= NFFT ? (p - NFFT) / HOP + 1 : 0;
    float2 acc = make_float2(0.f, 0.f);
    for (int f = f_lo; f <= f_hi; ++f)
    {
        const float2 c = frames[((size_t)src * T + f) * NFFT + (p - f * HOP)];
        acc.x += c.x;
        acc.y += c.y;
    }
    reinterpret_cast<float2 *>(out.p[blockIdx.z][src])[s] = acc;
}

} // namespace umx
