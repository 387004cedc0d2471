// wiener_kernels.h -- multichannel Wiener filter, one EM iteration (wiener.cpp:92-425).
//
// Reproduces the reference including its load-bearing quirks: the PSD is mean_c (Re y + Im y)^2
// (wiener.cpp:187-202) and sqrt(eps) I is added once per source (wiener.cpp:311-323,
// F6).  The initial estimate y_j = polar(mag_j, arg X) (wiener.cpp:96-109) is formed as
// mag_j * X/|X| (phase 0 where X = 0, like std::arg), never materialised in HBM.
//
//   pass 1  wiener_stats4_kernel  per (200-frame batch, bin), all sources: sum_f y y^H and sum_f v, in the reference's
//                                 batch structure (wiener.cpp:212-258)
//   pass 2  wiener_finish4_kernel R_j(b) = (sum of batch sums, in batch order) / (eps + sum v)
//   pass 3  wiener_apply_kernel   per (f, b): Cxx, closed-form 2x2 inverse (wiener.cpp:54-84), gains, y_j = G_j x,
//                                 * max_abs -> y [4][2][T][2049]   (single-track contexts; track-batched contexts fuse
//                                 this pass with the inverse STFT: wiener_istft.h)
// The target magnitudes arrive as MASKS [2][T][MAGP] (gemm_common.h): mag_j = mask_j x |X| (inference.cpp:175-183) is
// formed here, from the mixture bin that is in registers anyway.
// (Round 1's per-source statistics kernels -- one frame in flight per thread, the mixture read four times -- are gone.)
#pragma once
#include "fft4096.h"

namespace umx
{

struct WienerMags
{
    const float *m[4]; // the four targets' masks, [2][T][MAGP] each
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
    const size_t j0 = mask_index(0, T, f, b), j1 = mask_index(1, T, f, b);
    const float2 X0 = spec[i0], X1 = spec[i1];
    const float h0 = mix_magnitude(X0), h1 = mix_magnitude(X1);
    const float2 x0 = make_float2(X0.x / max_abs, X0.y / max_abs); // wiener.cpp:118-130
    const float2 x1 = make_float2(X1.x / max_abs, X1.y / max_abs);
    float v[4];
    float2 Rr[4][2][2];
    float2 C[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        const float2 ya = wiener_y0(mags.m[s][j0] * h0, X0, max_abs); // inference.cpp:175-183: mask x |X|
        const float2 yb = wiener_y0(mags.m[s][j1] * h1, X1, max_abs);
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

// ------------------------------------------------------------------------------------------------------------------
// The statistics pass.
//
// R_j is Hermitian BIT FOR BIT: R10 accumulates y1 conj(y0), whose real part is the same two products in the same
// order as R01's and whose imaginary part is its exact negation; R00 and R11 have an exactly zero imaginary part
// (-ab + ba).  So four floats per (source, bin) carry everything: {R00, Re R01, Im R01, R11}.
//
//   wiener_stats4_kernel   one thread = one bin x one batch of the reference (wiener.cpp:212-258) x ALL FOUR sources,
//                          frames accumulated in the reference's order (same bits as the per-source kernel above):
//                          the mixture spectrogram and its unit phasor are read / formed once instead of four times,
//                          and the loads of the NEXT eight frames are in flight while eight are accumulated (the
//                          per-source kernel had one frame in flight per thread and ~1.5 waves per SIMD to hide the
//                          latency behind: 1.5 TB/s).  64-thread workgroups so that the 13 x 33 of them cover the chip.
//   wiener_finish4_kernel  R_j(b) = (batch sums in batch order) / (eps + sum v)  -> Rc [4][2049][4]
//   wiener_istft_kernel    (wiener_istft.h) gains + filter + inverse STFT frame in one pass: y never goes to HBM.
constexpr int WIENER_CHUNK = WIENER_BATCH; // frames per thread of wiener_stats4_kernel (a divisor of WIENER_BATCH)
static_assert(WIENER_BATCH % WIENER_CHUNK == 0, "chunks must not straddle the reference's batches");
constexpr int WIENER_PF = 8;               // frames per prefetch group

template <int NS> struct WienerFrame // what one frame contributes to one bin: mixture (2 channels) and NS x 2 magnitudes
{
    float2 X0, X1;
    float m0[NS], m1[NS];
};
template <int NS>
__device__ __forceinline__ void wiener_frame_load(WienerFrame<NS> &w, const float2 *__restrict__ spec, const float *const (&mag)[NS],
                                                  int T, int f, int b)
{
    const size_t i0 = ((size_t)0 * T + f) * NBINS + b, i1 = ((size_t)1 * T + f) * NBINS + b;
    const size_t j0 = mask_index(0, T, f, b), j1 = mask_index(1, T, f, b);
    w.X0 = spec[i0];
    w.X1 = spec[i1];
#pragma unroll
    for (int s = 0; s < NS; ++s)
    {
        w.m0[s] = mag[s][j0]; // masks: x |X| in accumulate()
        w.m1[s] = mag[s][j1];
    }
}

// grid (ceil(B/64), nchunk, 4 / NS), 64 threads; a thread handles the NS sources NS z .. NS z + NS - 1 (the more sources
// per thread, the fewer times the mixture is read and its phasor formed; the fewer, the more waves to spread over the
// chip: the accumulation is arithmetic- and latency-bound per wave).
// part: [nchunk][4 sources][5][2049] = R00, Re R01, Im R01, R11, sum v (bin fastest)
// Lanes: blockIdx.y = entry * nchunk + chunk; spec, mags, maxabs_bits and part are lane 0's (WienerStrides apart per lane).
struct WienerStrides
{
    size_t spec, mag, part, rc, r8, frames, y; // elements between consecutive track lanes
};
template <int NS>
__global__ __launch_bounds__(64) void wiener_stats4_kernel(const float2 *__restrict__ spec, WienerMags mags, int T,
                                                           const unsigned *__restrict__ maxabs_bits,
                                                           float *__restrict__ part, LaneSet lanes, WienerStrides ls)
{
    const int nchunk_all = (T + WIENER_CHUNK - 1) / WIENER_CHUNK;
    const int b = min(blockIdx.x * 64 + threadIdx.x, NBINS - 1), chunk = blockIdx.y % nchunk_all; // surplus lanes repeat the last bin
    const int s0 = NS * blockIdx.z;
    {
        const int ln = lanes.id[blockIdx.y / nchunk_all];
        spec += (size_t)ln * ls.spec;
        part += (size_t)ln * ls.part;
        maxabs_bits += ln;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            mags.m[s] += (size_t)ln * ls.mag;
    }
    const float max_abs = wiener_max_abs(maxabs_bits), rmax = 1.0f / max_abs;
    const int f0 = chunk * WIENER_CHUNK, f1 = min(T, f0 + WIENER_CHUNK);
    const float *mag[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        mag[s] = NS == 4 ? mags.m[s] : (s0 + s == 0 ? mags.m[0] : s0 + s == 1 ? mags.m[1] : s0 + s == 2 ? mags.m[2] : mags.m[3]);
    float r00[NS], r01x[NS], r01y[NS], r11[NS], wsum[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        r00[s] = r01x[s] = r01y[s] = r11[s] = wsum[s] = 0.f;
    auto accumulate = [&](const WienerFrame<NS> &w) {
        const float2 p0 = unit_phasor(w.X0), p1 = unit_phasor(w.X1);
        const float h0 = mix_magnitude(w.X0), h1 = mix_magnitude(w.X1);
#pragma unroll
        for (int s = 0; s < NS; ++s)
        {
            const float t0 = w.m0[s] * h0, t1 = w.m1[s] * h1; // target magnitude = mask x |X| (inference.cpp:175-183)
            // wiener_y0 with the division by max_abs as an exact 3-instruction quotient (div_by, common.h)
            const float2 y0 = make_float2(div_by(t0 * p0.x, max_abs, rmax), div_by(t0 * p0.y, max_abs, rmax));
            const float2 y1 = make_float2(div_by(t1 * p1.x, max_abs, rmax), div_by(t1 * p1.y, max_abs, rmax));
            // v = 1/2 sum_c (Re + Im)^2   wiener.cpp:187-202 (F5)
            const float ra = y0.x + y0.y, rb = y1.x + y1.y;
            float sum = 0.f;
            sum += (ra * ra) + (0.f * 0.f);
            sum += (rb * rb) + (0.f * 0.f);
            wsum[s] += sum / 2;
            // calculateCovariance wiener.cpp:435-478: a * conj(b); only the independent entries
            const float2 q00 = cmul(y0, cconj(y0)), q01 = cmul(y0, cconj(y1)), q11 = cmul(y1, cconj(y1));
            r00[s] += (0.f + q00.x);
            r01x[s] += (0.f + q01.x);
            r01y[s] += (0.f + q01.y);
            r11[s] += (0.f + q11.x);
        }
    };
    // a ring of PF frames in registers: a slot is refilled (frame + PF) as soon as its frame has been accumulated, so PF
    // frames' loads are in flight at all times
    constexpr int PF = NS == 1 ? 16 : NS == 2 ? 12 : WIENER_PF;
    WienerFrame<NS> ringf[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k)
        wiener_frame_load<NS>(ringf[k], spec, mag, T, min(f0 + k, f1 - 1), b);
    for (int f = f0; f < f1; f += PF)
    {
#pragma unroll
        for (int k = 0; k < PF; ++k)
        {
            if (f + k < f1)
                accumulate(ringf[k]);
            wiener_frame_load<NS>(ringf[k], spec, mag, T, min(f + PF + k, f1 - 1), b);
        }
    }
    if (blockIdx.x * 64 + threadIdx.x >= NBINS)
        return;
#pragma unroll
    for (int s = 0; s < NS; ++s)
    {
        float *o = part + ((size_t)(chunk * 4 + s0 + s) * 5) * NBINS + b;
        o[0] = r00[s];
        o[NBINS] = r01x[s];
        o[2 * NBINS] = r01y[s];
        o[3 * NBINS] = r11[s];
        o[4 * NBINS] = wsum[s];
    }
}

// grid (ceil(B/256), 4).  Rc: [4][2049][4] = {R00, Re R01, Im R01, R11}; R8 (optional): [4][2049][8]
__global__ __launch_bounds__(256) void wiener_finish4_kernel(const float *__restrict__ part, int T, float *__restrict__ Rc,
                                                             float *__restrict__ R8, LaneSet lanes, WienerStrides ls)
{
    const int b = blockIdx.x * 256 + threadIdx.x, src = blockIdx.y;
    if (b >= NBINS)
        return;
    {
        const int ln = lanes.id[blockIdx.z]; // grid (ceil(B/256), 4, lanes)
        part += (size_t)ln * ls.part;
        Rc += (size_t)ln * ls.rc;
        if (R8)
            R8 += (size_t)ln * ls.r8;
    }
    constexpr int CPB = WIENER_BATCH / WIENER_CHUNK;
    const int nchunk = (T + WIENER_CHUNK - 1) / WIENER_CHUNK;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float weight = WIENER_EPS; // wiener.cpp:210
    for (int c0 = 0; c0 < nchunk; c0 += CPB)
    {
        float bs[5] = {0.f, 0.f, 0.f, 0.f, 0.f}; // one batch of the reference = CPB chunks, in frame order
        for (int c = c0; c < min(nchunk, c0 + CPB); ++c)
        {
            const float *p = part + ((size_t)(c * 4 + src) * 5) * NBINS + b;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                bs[i] += p[(size_t)i * NBINS];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] += bs[i]; // wiener.cpp:243  R += batch sum
        weight += bs[4];     // wiener.cpp:247-253
    }
    const float4 r = make_float4(acc[0] / weight, acc[1] / weight, acc[2] / weight, acc[3] / weight); // wiener.cpp:259-269
    *reinterpret_cast<float4 *>(Rc + ((size_t)src * NBINS + b) * 4) = r;
    if (R8) // the eight-float form wiener_apply_kernel reads: R00, R01, R10 = conj(R01), R11
    {
        float4 *o = reinterpret_cast<float4 *>(R8 + ((size_t)src * NBINS + b) * 8);
        o[0] = make_float4(r.x, 0.f, r.y, r.z);
        o[1] = make_float4(r.y, -r.z, r.w, 0.f);
    }
}

// The per-bin part of wiener_apply_kernel, split in two so that the fused kernel can hold many bins in registers:
// WienerBin = everything that does not depend on the source whose output is being formed.
struct WienerBin
{
    // inverse of Cxx (wiener.cpp:54-84).  Cxx = sum_s (sqrt(eps) I + v_s R_s) is Hermitian bit for bit like the R_s
    // (real diagonal, C10 = conj(C01) exactly), its determinant is exactly real, and so its inverse is Hermitian too:
    // four floats {Ci00, Re Ci01, Im Ci01, Ci11}
    float ci00, ci01x, ci01y, ci11;
    float2 x0, x1; // mixture / max_abs
    float v[4];    // source PSDs
};

__device__ __forceinline__ void wiener_expand(float4 rc, float2 (&Rr)[2][2])
{
    Rr[0][0] = make_float2(rc.x, 0.f);
    Rr[0][1] = make_float2(rc.y, rc.z);
    Rr[1][0] = make_float2(rc.y, -rc.z);
    Rr[1][1] = make_float2(rc.w, 0.f);
}

// Both functions are the generic complex 2x2 arithmetic of wiener_apply_kernel with the structural zeros taken out: R_j,
// Cxx and its inverse are Hermitian with exactly real diagonals (above), so every product with an exact zero and every
// sum with one is dropped -- x * 0 is a zero and y + 0 is y, so the VALUES are those of the generic form (only the sign
// of a zero can differ), at about half the instructions; the remaining operations keep the generic form's order and
// rounding.  tests/test_gpu_parity.py compares the fused kernel against wiener_apply_kernel bit for bit.
__device__ __forceinline__ void wiener_bin_setup(float2 X0, float2 X1, const float (&m0)[4], const float (&m1)[4],
                                                 const float4 (&rc)[4], float max_abs, float rmax, WienerBin &w)
{
    const float reg = sqrtf(WIENER_EPS); // wiener.cpp:165
    w.x0 = make_float2(div_by(X0.x, max_abs, rmax), div_by(X0.y, max_abs, rmax)); // wiener.cpp:118-130
    w.x1 = make_float2(div_by(X1.x, max_abs, rmax), div_by(X1.y, max_abs, rmax));
    const float2 p0 = unit_phasor(X0), p1 = unit_phasor(X1);
    float c00 = 0.f, c11 = 0.f, c01x = 0.f, c01y = 0.f; // Cxx = [[c00, c01], [conj(c01), c11]]
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        const float2 ya = make_float2(div_by(m0[s] * p0.x, max_abs, rmax), div_by(m0[s] * p0.y, max_abs, rmax));
        const float2 yb = make_float2(div_by(m1[s] * p1.x, max_abs, rmax), div_by(m1[s] * p1.y, max_abs, rmax));
        const float ra = ya.x + ya.y, rb = yb.x + yb.y;
        float sum = 0.f;
        sum += (ra * ra) + (0.f * 0.f);
        sum += (rb * rb) + (0.f * 0.f);
        const float v = sum / 2;
        w.v[s] = v;
        // wiener.cpp:307-325: Cxx += reg(c1,c2) + v * R   (F6: once per source)
        c00 += reg + v * rc[s].x;
        c01x += v * rc[s].y;
        c01y += v * rc[s].z;
        c11 += reg + v * rc[s].w;
    }
    // invert4D wiener.cpp:54-84: det = c00 c11 - c01 conj(c01) is real; 1/det is formed as det / |det|^2 like the generic form
    const float det = c00 * c11 - (c01x * c01x + c01y * c01y);
    const float idet = det / (det * det);
    w.ci00 = idet * c11;
    w.ci01x = -idet * c01x;
    w.ci01y = -idet * c01y;
    w.ci11 = idet * c00;
}

// y_s = G_s x * max_abs for one source (wiener.cpp:343-400): G = (R Cxx^-1) v, R = [[a, b], [conj(b), d]], Cxx^-1 = [[p, q], [conj(q), r]]
__device__ __forceinline__ void wiener_bin_apply(const WienerBin &w, int s, float4 rc, float max_abs, float2 (&o)[2])
{
    const float vs = s == 0 ? w.v[0] : s == 1 ? w.v[1] : s == 2 ? w.v[2] : w.v[3]; // selects: w stays in registers
    const float a = rc.x, bx = rc.y, by = rc.z, d = rc.w, p = w.ci00, qx = w.ci01x, qy = w.ci01y, r = w.ci11;
    float2 g[2][2];
    // g00 = a p + b conj(q)
    g[0][0] = make_float2(a * p + (bx * qx + by * qy), by * qx - bx * qy);
    // g01 = a q + b r
    g[0][1] = make_float2(a * qx + bx * r, a * qy + by * r);
    // g10 = conj(b) p + d conj(q)
    g[1][0] = make_float2(bx * p + d * qx, -(by * p + d * qy));
    // g11 = conj(b) q + d r
    g[1][1] = make_float2((bx * qx + by * qy) + d * r, bx * qy - by * qx);
#pragma unroll
    for (int c1 = 0; c1 < 2; ++c1)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
            g[c1][c2] = make_float2(g[c1][c2].x * vs, g[c1][c2].y * vs);
    o[0] = make_float2(0.f, 0.f);
    o[1] = make_float2(0.f, 0.f);
#pragma unroll
    for (int c1 = 0; c1 < 2; ++c1) // wiener.cpp:381-400
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
            o[c2] = cadd(o[c2], cmul(g[c2][c1], c1 == 0 ? w.x0 : w.x1));
    o[0] = make_float2(o[0].x * max_abs, o[0].y * max_abs);
    o[1] = make_float2(o[1].x * max_abs, o[1].y * max_abs);
}

// "no Wiener" configuration (BASELINE config 2): y_j = mag_j * exp(i arg X)  (wiener.cpp:96-109 only)
__global__ __launch_bounds__(256) void mixphase_kernel(const float2 *__restrict__ spec, WienerMags mags, int T,
                                                       float2 *__restrict__ y)
{
    const size_t n = (size_t)2 * T * NBINS;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float2 X = spec[i];
    const float2 ph = unit_phasor(X);
    const float h = mix_magnitude(X);
    const size_t j = (i / NBINS) * MAGP + i % NBINS; // (c, f) row of the mask planes
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        const float m = mags.m[s][j] * h; // inference.cpp:175-183
        y[(size_t)s * n + i] = make_float2(m * ph.x, m * ph.y);
    }
}

} // namespace umx
