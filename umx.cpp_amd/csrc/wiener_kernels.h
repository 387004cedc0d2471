// wiener_kernels.h -- multichannel Wiener filter, one EM iteration (wiener.cpp:92-425).
//
// Reproduces the reference including its load-bearing quirks: the PSD is mean_c (Re y + Im y)^2
// (wiener.cpp:187-202) and sqrt(eps) I is added once per source (wiener.cpp:311-323,
// F6).  The initial estimate y_j = polar(mag_j, arg X) (wiener.cpp:96-109) is formed as
// mag_j * X/|X| (phase 0 where X = 0, like std::arg), never materialised in HBM.
//
//   pass 1  wiener_stats_kernel   per (source, 200-frame batch, bin): sum_f y y^H and sum_f v, in
//                                 the reference's batch structure (wiener.cpp:212-258)
//   pass 2  wiener_finish_kernel  R_j(b) = (sum of batch sums, in batch order) / (eps + sum v)
//   pass 3  wiener_apply_kernel   per (f, b): Cxx, closed-form 2x2 inverse (wiener.cpp:54-84),
//                                 gains, y_j = G_j x, * max_abs            -> y [4][2][T][2049]
// All three are HBM-streaming kernels: algorithmic bytes per 60 s segment
//   pass 1: 4 x (84.7 spec + 42.4 mag) = 508 MB read;  pass 3: 84.7 + 169.4 read, 338.8 written.
#pragma once
#include "fft4096.h"

namespace umx
{

struct WienerMags
{
    const float *m[4]; // [2][T][2049] each
};

__device__ __forceinline__ float2 unit_phasor(float2 x)
{
    const float a = sqrtf(x.x * x.x + x.y * x.y);
    return a > 0.f ? make_float2(x.x / a, x.y / a) : make_float2(1.f, 0.f); // arg(0) = 0
}

__device__ __forceinline__ float wiener_max_abs(const unsigned *maxabs_bits)
{
    return fmaxf(1.0f, __uint_as_float(*maxabs_bits) / WIENER_SCALE); // wiener.cpp:51
}

// y_j(c) scaled down by max_abs exactly like wiener.cpp:133-146 (component / max_abs)
__device__ __forceinline__ float2 wiener_y0(float mag, float2 x, float max_abs)
{
    const float2 ph = unit_phasor(x);
    return make_float2((mag * ph.x) / max_abs, (mag * ph.y) / max_abs);
}

// grid (ceil(B/256), nbatch, 4).  part: [4][nbatch][2049][9] = Re/Im of R00 R01 R10 R11, sum v
__global__ __launch_bounds__(256) void wiener_stats_kernel(const float2 *__restrict__ spec, WienerMags mags,
                                                           int T, const unsigned *__restrict__ maxabs_bits,
                                                           float *__restrict__ part, int nbatch)
{
    const int b = blockIdx.x * 256 + threadIdx.x, batch = blockIdx.y, src = blockIdx.z;
    if (b >= NBINS)
        return;
    const float max_abs = wiener_max_abs(maxabs_bits);
    const float *mag = mags.m[src];
    const int f0 = batch * WIENER_BATCH, f1 = min(T, f0 + WIENER_BATCH);
    float2 r[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
    float wsum = 0.f;
    for (int f = f0; f < f1; ++f)
    {
        const size_t i0 = ((size_t)0 * T + f) * NBINS + b, i1 = ((size_t)1 * T + f) * NBINS + b;
        float2 y[2];
        y[0] = wiener_y0(mag[i0], spec[i0], max_abs);
        y[1] = wiener_y0(mag[i1], spec[i1], max_abs);
        // v = 1/2 sum_c (Re + Im)^2   wiener.cpp:187-202 (F5)
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
        {
            const float re = y[c].x + y[c].y;
            sum += (re * re) + (0.f * 0.f);
        }
        wsum += sum / 2;
#pragma unroll
        for (int c1 = 0; c1 < 2; ++c1)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
            { // calculateCovariance wiener.cpp:435-478: a * conj(b)
                const float2 a = y[c1], bb = cconj(y[c2]);
                const float2 pr = cmul(a, bb);
                r[c1][c2].x += (0.f + pr.x);
                r[c1][c2].y += (0.f + pr.y);
            }
    }
    float *o = part + (((size_t)src * nbatch + batch) * NBINS + b) * 9;
    o[0] = r[0][0].x; o[1] = r[0][0].y; o[2] = r[0][1].x; o[3] = r[0][1].y;
    o[4] = r[1][0].x; o[5] = r[1][0].y; o[6] = r[1][1].x; o[7] = r[1][1].y;
    o[8] = wsum;
}

// grid (ceil(B/256), 4).  R: [4][2049][8]
__global__ __launch_bounds__(256) void wiener_finish_kernel(const float *__restrict__ part, int nbatch,
                                                            float *__restrict__ R)
{
    const int b = blockIdx.x * 256 + threadIdx.x, src = blockIdx.y;
    if (b >= NBINS)
        return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float weight = WIENER_EPS; // wiener.cpp:210
    for (int k = 0; k < nbatch; ++k)
    {
        const float *p = part + (((size_t)src * nbatch + k) * NBINS + b) * 9;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] += p[i]; // wiener.cpp:243  R += batch sum
        weight += p[8];     // wiener.cpp:247-253 (batch partial sums: rounding-order difference only)
    }
    float *o = R + ((size_t)src * NBINS + b) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        o[i] = acc[i] / weight; // wiener.cpp:259-269
}

// grid (ceil(B/256), T).  y: [4][2][T][2049] complex
__global__ __launch_bounds__(256) void wiener_apply_kernel(const float2 *__restrict__ spec, WienerMags mags,
                                                           int T, const unsigned *__restrict__ maxabs_bits,
                                                           const float *__restrict__ R, float2 *__restrict__ y)
{
    const int b = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (b >= NBINS)
        return;
    const float max_abs = wiener_max_abs(maxabs_bits);
    const float reg = sqrtf(WIENER_EPS); // wiener.cpp:165
    const size_t i0 = ((size_t)0 * T + f) * NBINS + b, i1 = ((size_t)1 * T + f) * NBINS + b;
    const float2 X0 = spec[i0], X1 = spec[i1];
    const float2 x0 = make_float2(X0.x / max_abs, X0.y / max_abs); // wiener.cpp:118-130
    const float2 x1 = make_float2(X1.x / max_abs, X1.y / max_abs);
    float v[4];
    float2 Rr[4][2][2];
    float2 C[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        const float2 ya = wiener_y0(mags.m[s][i0], X0, max_abs);
        const float2 yb = wiener_y0(mags.m[s][i1], X1, max_abs);
        const float ra = ya.x + ya.y, rb = yb.x + yb.y;
        float sum = 0.f;
        sum += (ra * ra) + (0.f * 0.f);
        sum += (rb * rb) + (0.f * 0.f);
        v[s] = sum / 2;
        const float4 *rp = reinterpret_cast<const float4 *>(R + ((size_t)s * NBINS + b) * 8);
        const float4 r01 = rp[0], r23 = rp[1];
        Rr[s][0][0] = make_float2(r01.x, r01.y);
        Rr[s][0][1] = make_float2(r01.z, r01.w);
        Rr[s][1][0] = make_float2(r23.x, r23.y);
        Rr[s][1][1] = make_float2(r23.z, r23.w);
#pragma unroll
        for (int c1 = 0; c1 < 2; ++c1)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
            { // wiener.cpp:307-325: Cxx += reg(c1,c2) + v * R   (F6: once per source)
                const float2 term = make_float2((c1 == c2 ? reg : 0.f) + v[s] * Rr[s][c1][c2].x,
                                                0.f + v[s] * Rr[s][c1][c2].y);
                C[c1][c2].x += term.x;
                C[c1][c2].y += term.y;
            }
    }
    // invert4D wiener.cpp:54-84
    const float2 det = csub(cmul(C[0][0], C[1][1]), cmul(C[0][1], C[1][0]));
    const float nrm = det.x * det.x + det.y * det.y;
    const float2 invDet = make_float2(det.x / nrm, -det.y / nrm);
    const float2 nInv = make_float2(-invDet.x, -invDet.y);
    float2 Ci[2][2];
    Ci[0][0] = cmul(invDet, C[1][1]);
    Ci[0][1] = cmul(nInv, C[0][1]);
    Ci[1][0] = cmul(nInv, C[1][0]);
    Ci[1][1] = cmul(invDet, C[0][0]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        float2 g[2][2];
#pragma unroll
        for (int c1 = 0; c1 < 2; ++c1)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
            { // wiener.cpp:343-376
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int c3 = 0; c3 < 2; ++c3)
                    acc = cadd(acc, cmul(Rr[s][c1][c3], Ci[c3][c2]));
                g[c1][c2] = make_float2(acc.x * v[s], acc.y * v[s]);
            }
        float2 o[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int c1 = 0; c1 < 2; ++c1) // wiener.cpp:381-400
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
                o[c2] = cadd(o[c2], cmul(g[c2][c1], c1 == 0 ? x0 : x1));
        y[(((size_t)s * 2 + 0) * T + f) * NBINS + b] = make_float2(o[0].x * max_abs, o[0].y * max_abs);
        y[(((size_t)s * 2 + 1) * T + f) * NBINS + b] = make_float2(o[1].x * max_abs, o[1].y * max_abs);
    }
}

// "no Wiener" configuration (BASELINE config 2): y_j = mag_j * exp(i arg X)  (wiener.cpp:96-109 only)
__global__ __launch_bounds__(256) void mixphase_kernel(const float2 *__restrict__ spec, WienerMags mags, int T,
                                                       float2 *__restrict__ y)
{
    const size_t n = (size_t)2 * T * NBINS;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float2 ph = unit_phasor(spec[i]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        const float m = mags.m[s][i];
        y[(size_t)s * n + i] = make_float2(m * ph.x, m * ph.y);
    }
}

} // namespace umx
