// fft4096.h -- 4096-point complex fp32 FFT for one 256-thread workgroup (gfx950).
//
// Replaces Eigen::FFT<float> (kissfft) as used by dsp.cpp:130-139,226-227,243-244.  One stereo
// frame is ONE complex transform: z = L + iR, so the two real 4096-point FFTs the reference
// runs per frame (one per channel) become a single radix-16 x 16 x 16 Stockham transform whose
// spectra are separated by conjugate symmetry afterwards.
//
// Each of the 256 threads holds one radix-16 butterfly in registers per pass; the two
// inter-pass exchanges go through a padded LDS buffer (index i -> i + i/16, which makes the
// stride-16 writes of pass 0 and the stride-272 writes of pass 1 conflict-free for
// ds_write_b64 and keeps every read a unit-stride ds_read_b64).  Twiddles come from two small
// tables in global memory laid out [r][lane] so that a wave reads them coalesced:
//   tw1[r][k]  = exp(-2 pi i r k / 256),  k < 16      (pass 1)
//   tw2[r][j]  = exp(-2 pi i r j / 4096), j < 256     (pass 2)
#pragma once
#include "common.h"

namespace umx
{

constexpr int FFT_LDS_ELEMS = NFFT + NFFT / 16; // padded float2 count (34,816 B)

__device__ __forceinline__ int fft_pad(int i) { return i + (i >> 4); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// the transform's own twiddle products (not the reference's arithmetic: its FFT is kissfft): two multiplies and two fused multiply-adds
// instead of four multiplies, an add and a subtract -- the engine is built with -ffp-contract=off, so the fusion is spelled out
__device__ __forceinline__ float2 cmul_tw(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// multiply by W4 = exp(-+ i pi/2): -i forward, +i inverse
template <bool INV> __device__ __forceinline__ float2 mul_w4(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// 16-point DFT in registers, 4x4 decomposition: Y[4*q1+q0] = sum_r v[r] W16^(r*(4*q1+q0))
template <bool INV> __device__ __forceinline__ void fft16(float2 (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f; // cos(pi/8)
    constexpr float S1 = 0.38268343236508977f; // sin(pi/8)
    constexpr float R2 = 0.70710678118654752f;
    constexpr float SG = INV ? 1.0f : -1.0f; // sign of the imaginary part of W16^m
    float2 a[4][4];                          // a[r0][q0]
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0)
    {
        float2 x0 = v[r0], x1 = v[4 + r0], x2 = v[8 + r0], x3 = v[12 + r0];
        float2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
        float2 rot = mul_w4<INV>(d13);
        a[r0][0] = cadd(s02, s13);
        a[r0][1] = cadd(d02, rot);
        a[r0][2] = csub(s02, s13);
        a[r0][3] = csub(d02, rot);
    }
    // inner twiddles W16^(r0*q0): m = 1,2,3 (r0=1), 2,4,6 (r0=2), 3,6,9 (r0=3)
    const float2 w1 = make_float2(C1, SG * S1), w2 = make_float2(R2, SG * R2),
                 w3 = make_float2(S1, SG * C1), w6 = make_float2(-R2, SG * R2),
                 w9 = make_float2(-C1, -SG * S1);
    a[1][1] = cmul_tw(a[1][1], w1);
    a[1][2] = cmul_tw(a[1][2], w2);
    a[1][3] = cmul_tw(a[1][3], w3);
    a[2][1] = cmul_tw(a[2][1], w2);
    a[2][2] = mul_w4<INV>(a[2][2]); // W16^4
    a[2][3] = cmul_tw(a[2][3], w6);
    a[3][1] = cmul_tw(a[3][1], w3);
    a[3][2] = cmul_tw(a[3][2], w6);
    a[3][3] = cmul_tw(a[3][3], w9);
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0)
    {
        float2 x0 = a[0][q0], x1 = a[1][q0], x2 = a[2][q0], x3 = a[3][q0];
        float2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
        float2 rot = mul_w4<INV>(d13);
        v[q0] = cadd(s02, s13);
        v[4 + q0] = cadd(d02, rot);
        v[8 + q0] = csub(s02, s13);
        v[12 + q0] = csub(d02, rot);
    }
}

// In: v[r] = input[j + 256*r] for this thread j = threadIdx.x (256 threads).
// Out: buf[fft_pad(k)] = transform bin k, natural order, visible after the final barrier.
// Unscaled in both directions (dsp.cpp:136 Unscaled flag).
// j: the thread's index inside the 256-thread group that owns buf (several groups of one workgroup may transform
// side by side, each in its own buffer; the barriers are workgroup-wide, so every group must make the same calls).
// KEEP: leave the transform in registers instead -- v[q] = bin j + 256 q, the thread's own sixteen outputs of the last pass -- for a caller
// whose next step is per element (the fused inverse STFT's weighting and overlap-add): no final store, no final barrier, no read back.
template <bool INV, bool KEEP = false>
__device__ __forceinline__ void fft4096(float2 (&v)[16], float2 *buf, const float2 *__restrict__ tw1,
                                        const float2 *__restrict__ tw2, int j)
{
    // pass 0 (Ns = 1): no twiddles
    fft16<INV>(v);
#pragma unroll
    for (int q = 0; q < 16; ++q)
        buf[fft_pad((j << 4) + q)] = v[q];
    __syncthreads();
    // pass 1 (Ns = 16)
    {
        const int k = j & 15;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] = buf[fft_pad(j + 256 * r)];
        __syncthreads();
#pragma unroll
        for (int r = 1; r < 16; ++r)
        {
            float2 w = tw1[r * 16 + k];
            v[r] = cmul_tw(v[r], INV ? cconj(w) : w);
        }
        fft16<INV>(v);
        const int base = ((j - k) << 4) + k;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            buf[fft_pad(base + 16 * q)] = v[q];
    }
    // pass 2's twiddles are requested in front of the barrier (v is dead here: the registers are free), so that they arrive while the
    // workgroup waits for its slowest wave instead of behind the fragment reads
    float2 w2[16];
#pragma unroll
    for (int r = 1; r < 16; ++r)
        w2[r] = tw2[r * 256 + j];
    __syncthreads();
    // pass 2 (Ns = 256): k = j, in-place per thread (reads and writes buf[j + 256*q])
    {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] = buf[fft_pad(j + 256 * r)];
#pragma unroll
        for (int r = 1; r < 16; ++r)
        {
            const float2 w = w2[r];
            v[r] = cmul_tw(v[r], INV ? cconj(w) : w);
        }
        fft16<INV>(v);
        if (!KEEP)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                buf[fft_pad(j + 256 * q)] = v[q];
    }
    if (!KEEP)
        __syncthreads();
}

template <bool INV>
__device__ __forceinline__ void fft4096(float2 (&v)[16], float2 *buf, const float2 *__restrict__ tw1,
                                        const float2 *__restrict__ tw2)
{
    fft4096<INV>(v, buf, tw1, tw2, (int)threadIdx.x);
}

} // namespace umx
