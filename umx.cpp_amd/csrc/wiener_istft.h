// wiener_istft.h -- Wiener gains + filter (wiener.cpp:270-425) fused with the inverse STFT frame (dsp.cpp:229-258):
// the filtered spectrograms y [4][2][T][2049] complex (339 MB per 60 s segment, written by one kernel and read back by
// the next) never touch HBM.
//
// One workgroup per frame, 1024 threads.
//   phase 1  thread t handles bins b = t + 1024 q (2049 bins, 2-3 per thread): loads the mixture (2 channels), the four
//            targets' masks (x 2 channels; x |X| = the target magnitudes) and the four R (16 bytes each), forms what does not depend on the output
//            source -- mixture over max_abs, the four PSDs, the inverse of Cxx (wiener.cpp:270-341) -- and then, for each
//            source, y_s = G_s x (wiener.cpp:343-400), written straight into the INPUT of that source's inverse FFT in
//            LDS: one complex 4096-point transform per source carries the left channel in its real and the right channel
//            in its imaginary part (stft_kernels.h), so bin b fills two slots:
//                in[b]        = L[b] + i R[b]                  b <= 2048
//                in[4096 - b] = conj(L[b]) + i conj(R[b])      0 < b < 2048
//   phase 2  the four 256-thread groups transform the four sources side by side (4 x 34 KB of LDS), apply the
//            reference's per-sample weight (dsp.cpp:248-256) and store the frames.
// 16 waves per CU hide the latencies of both phases (a 256-thread version holding nine bins per thread in registers ran
// at one wave per SIMD and was no faster than the two kernels it replaced).  Per frame: 98 KB of spectrogram /
// magnitudes and 131 KB of R (L2-resident) in, 131 KB of frames out, against 229 + 131 KB in and 131 + 131 KB out.
// WIENER = false is BASELINE config 2: y_s = mag_s * exp(i arg X) (wiener.cpp:96-109 only).
#pragma once
#include "stft_kernels.h"
#include "wiener_kernels.h"

namespace umx
{

// NSRC = sources per workgroup (4: one workgroup per frame as described above; 2 / 1: grid.y = 2 / 4 workgroups per frame,
// each repeating phase 1's source-independent part for its own bins -- the second reads hit the L2 -- in exchange for
// 2 / 4 workgroups per CU whose phases overlap)
// rc[s] for a run-time s without indexing the array (it stays in registers)
__device__ __forceinline__ float4 sel4(int s, const float4 (&r)[4])
{
    return make_float4(s == 0 ? r[0].x : s == 1 ? r[1].x : s == 2 ? r[2].x : r[3].x, s == 0 ? r[0].y : s == 1 ? r[1].y : s == 2 ? r[2].y : r[3].y,
                       s == 0 ? r[0].z : s == 1 ? r[1].z : s == 2 ? r[2].z : r[3].z, s == 0 ? r[0].w : s == 1 ? r[1].w : s == 2 ? r[2].w : r[3].w);
}

template <bool WIENER, int NSRC>
__global__ __launch_bounds__(256 * NSRC) void wiener_istft_kernel(const float2 *__restrict__ spec, WienerMags mags, int T,
                                                                  const unsigned *__restrict__ maxabs_bits,
                                                                  const float *__restrict__ Rc, const float *__restrict__ window,
                                                                  const float *__restrict__ nw, const float2 *__restrict__ tw1,
                                                                  const float2 *__restrict__ tw2, float2 *__restrict__ frames,
                                                                  float2 *__restrict__ y_dbg, LaneSet lanes, WienerStrides ls)
{
    extern __shared__ __attribute__((aligned(16))) float2 wi_buf[]; // [NSRC][FFT_LDS_ELEMS]
    constexpr int WI_THREADS = 256 * NSRC;
    {
        const int ln = lanes.id[blockIdx.z]; // grid (T, 4 / NSRC, lanes): the pointers are lane 0's
        spec += (size_t)ln * ls.spec;
        Rc += (size_t)ln * ls.rc;
        frames += (size_t)ln * ls.frames;
        maxabs_bits += ln;
        if (y_dbg)
            y_dbg += (size_t)ln * ls.y;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            mags.m[s] += (size_t)ln * ls.mag;
    }
    const int src0 = NSRC * blockIdx.y;
    const int f = blockIdx.x, tid = threadIdx.x;
    const float max_abs = WIENER ? wiener_max_abs(maxabs_bits) : 1.0f, rmax = 1.0f / max_abs;
#pragma unroll
    for (int q = 0; q < (NFFT / 2 + WI_THREADS) / WI_THREADS; ++q)
    {
        const int b = tid + WI_THREADS * q;
        if (b > NFFT / 2)
            break;
        const size_t i0 = ((size_t)0 * T + f) * NBINS + b, i1 = ((size_t)1 * T + f) * NBINS + b;
        const size_t j0 = mask_index(0, T, f, b), j1 = mask_index(1, T, f, b);
        const float2 X0 = spec[i0], X1 = spec[i1];
        const float h0 = mix_magnitude(X0), h1 = mix_magnitude(X1);
        float m0[4], m1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            m0[s] = mags.m[s][j0] * h0; // target magnitude = mask x |X| (inference.cpp:175-183)
            m1[s] = mags.m[s][j1] * h1;
        }
        WienerBin wb;
        float4 rc[4];
        float2 p0, p1;
        if (WIENER)
        {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                rc[s] = *reinterpret_cast<const float4 *>(Rc + ((size_t)s * NBINS + b) * 4);
            wiener_bin_setup(X0, X1, m0, m1, rc, max_abs, rmax, wb);
        }
        else
        {
            p0 = unit_phasor(X0);
            p1 = unit_phasor(X1);
        }
#pragma unroll
        for (int sl = 0; sl < NSRC; ++sl)
        {
            const int s = src0 + sl;
            float2 o[2];
            if (WIENER)
                wiener_bin_apply(wb, s, NSRC == 4 ? rc[sl] : sel4(s, rc), max_abs, o);
            else
            {
                const float ms0 = NSRC == 4 ? m0[sl] : (s == 0 ? m0[0] : s == 1 ? m0[1] : s == 2 ? m0[2] : m0[3]);
                const float ms1 = NSRC == 4 ? m1[sl] : (s == 0 ? m1[0] : s == 1 ? m1[1] : s == 2 ? m1[2] : m1[3]);
                o[0] = make_float2(ms0 * p0.x, ms0 * p0.y);
                o[1] = make_float2(ms1 * p1.x, ms1 * p1.y);
            }
            if (y_dbg)
            {
                y_dbg[(((size_t)s * 2 + 0) * T + f) * NBINS + b] = o[0];
                y_dbg[(((size_t)s * 2 + 1) * T + f) * NBINS + b] = o[1];
            }
            float2 a = o[0], bb = o[1];
            if (b == 0 || b == NFFT / 2) // a real inverse FFT ignores Im of DC / Nyquist
            {
                a.y = 0.f;
                bb.y = 0.f;
            }
            float2 *buf = wi_buf + sl * FFT_LDS_ELEMS;
            buf[fft_pad(b)] = make_float2(a.x - bb.y, a.y + bb.x); // a + i b
            if (b > 0 && b < NFFT / 2)
                buf[fft_pad(NFFT - b)] = make_float2(a.x + bb.y, bb.x - a.y); // conj(a) + i conj(b)
        }
    }
    __syncthreads();
    const int g = tid >> 8, j = tid & 255; // source (of this workgroup's), thread of its transform
    float2 *buf = wi_buf + g * FFT_LDS_ELEMS;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        v[r] = buf[fft_pad(j + 256 * r)];
    __syncthreads();
    fft4096<true>(v, buf, tw1, tw2, j);
    float2 *dst = frames + ((size_t)(src0 + g) * T + f) * NFFT;
    const size_t start = (size_t)f * HOP;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        const int i = j + 256 * r;
        const float2 z = buf[fft_pad(i)];
        const float w = window[i];
        const float den = nw[start + i] + 1e-8f;
        dst[i] = make_float2(z.x * w * 1.0f / float(NFFT) / den, z.y * w * 1.0f / float(NFFT) / den); // dsp.cpp:248-256
    }
}

} // namespace umx
