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
//   pass 3  wiener_apply_kernel   per (f, b): Cxx, closed-form 2x2 inverse (wiener.cpp:54-84), y_j = v_j R_j (Cxx^-1 x)
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

__device__ __forceinline__ float wiener_max_abs(const unsigned *maxabs_bits)
{
    return fmaxf(1.0f, __uint_as_float(*maxabs_bits) / WIENER_SCALE); // wiener.cpp:51
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
constexpr int WIENER_STATS_NS = 4;         // sources per thread of wiener_stats4_kernel (round 5: 2 -> 4, the mixture is read once: 2.10 -> 1.56 ms per 32-lane launch)

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
    size_t spec, mag, part, rc, frames, y; // elements between consecutive track lanes
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
            // y_j(c) = mag * phasor / max_abs (wiener.cpp:133-146), the division as an exact 3-instruction quotient (div_by, common.h)
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

// grid (ceil(B/256), 4).  Rc: [4][2049][4] = {R00, Re R01, Im R01, R11}
__global__ __launch_bounds__(256) void wiener_finish4_kernel(const float *__restrict__ part, int T, float *__restrict__ Rc,
                                                             LaneSet lanes, WienerStrides ls)
{
    const int b = blockIdx.x * 256 + threadIdx.x, src = blockIdx.y;
    if (b >= NBINS)
        return;
    {
        const int ln = lanes.id[blockIdx.z]; // grid (ceil(B/256), 4, lanes)
        part += (size_t)ln * ls.part;
        Rc += (size_t)ln * ls.rc;
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
}

// The per-bin arithmetic of the filter, split in two so that the fused kernel (wiener_istft.h) can hold its bins in registers:
// WienerBin = everything that does not depend on the source whose output is being formed.
//
// y_j = G_j x with G_j = v_j R_j Cxx^-1 (wiener.cpp:343-400) is formed as  y_j = v_j * (R_j * (Cxx^-1 x)):  the reference builds the
// 2 x 2 complex gain of every source first (two matrix products and a scaling per source and bin) and then applies it; the product
// Cxx^-1 x does not depend on the source, so one matrix-vector product per bin and one per source give the same y_j with 25 instead
// of 73 operations per source and bin -- in a kernel bound by its vector-instruction count (wiener_istft.h: -12 %).  The two forms
// differ by rounding only (the sums are reassociated: measured against the oracle, which keeps the reference's order, in
// tests/test_gpu_parity.py).  Everything up to Cxx^-1 keeps the reference's operations and order.
struct WienerBin
{
    float2 t0, t1; // Cxx^-1 x, x = mixture / max_abs (wiener.cpp:118-130)
    float v[4];    // source PSDs
};

// R_j, Cxx and its inverse are Hermitian with exactly real diagonals (the statistics pass below; Cxx = sum_s (sqrt(eps) I + v_s R_s);
// its determinant is exactly real): four floats {M00, Re M01, Im M01, M11} carry each, and products with the structural zeros are
// not formed.
__device__ __forceinline__ void wiener_bin_setup(float2 X0, float2 X1, const float (&m0)[4], const float (&m1)[4],
                                                 const float4 (&rc)[4], float max_abs, float rmax, WienerBin &w)
{
    const float reg = sqrtf(WIENER_EPS); // wiener.cpp:165
    const float2 x0 = make_float2(div_by(X0.x, max_abs, rmax), div_by(X0.y, max_abs, rmax)); // wiener.cpp:118-130
    const float2 x1 = make_float2(div_by(X1.x, max_abs, rmax), div_by(X1.y, max_abs, rmax));
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
    // invert4D wiener.cpp:54-84: det = c00 c11 - c01 conj(c01) is real; 1/det is formed as det / |det|^2 like the reference's form
    const float det = c00 * c11 - (c01x * c01x + c01y * c01y);
    const float idet = det / (det * det);
    const float p = idet * c11, qx = -idet * c01x, qy = -idet * c01y, r = idet * c00; // Cxx^-1 = [[p, q], [conj(q), r]]
    // t = Cxx^-1 x
    w.t0 = make_float2(p * x0.x + (qx * x1.x - qy * x1.y), p * x0.y + (qx * x1.y + qy * x1.x));
    w.t1 = make_float2((qx * x0.x + qy * x0.y) + r * x1.x, (qx * x0.y - qy * x0.x) + r * x1.y);
}

// y_s = v_s R_s t * max_abs for one source: R = [[a, b], [conj(b), d]]
__device__ __forceinline__ void wiener_bin_apply(const WienerBin &w, int s, float4 rc, float max_abs, float2 (&o)[2])
{
    const float vs = s == 0 ? w.v[0] : s == 1 ? w.v[1] : s == 2 ? w.v[2] : w.v[3]; // selects: w stays in registers
    const float a = rc.x, bx = rc.y, by = rc.z, d = rc.w;
    const float2 t0 = w.t0, t1 = w.t1;
    const float2 u0 = make_float2(a * t0.x + (bx * t1.x - by * t1.y), a * t0.y + (bx * t1.y + by * t1.x));
    const float2 u1 = make_float2((bx * t0.x + by * t0.y) + d * t1.x, (bx * t0.y - by * t0.x) + d * t1.y);
    const float g = vs * max_abs; // wiener.cpp:408-422 undoes the scaling of wiener.cpp:115-146
    o[0] = make_float2(u0.x * g, u0.y * g);
    o[1] = make_float2(u1.x * g, u1.y * g);
}

// The separate filter pass of single-track contexts (track-batched contexts: wiener_istft.h).  R: the four-float form.
// grid (ceil(B/256), T).  y: [4][2][T][2049] complex
__global__ __launch_bounds__(256) void wiener_apply_kernel(const float2 *__restrict__ spec, WienerMags mags,
                                                           int T, const unsigned *__restrict__ maxabs_bits,
                                                           const float *__restrict__ R, float2 *__restrict__ y)
{
    const int b = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (b >= NBINS)
        return;
    const float max_abs = wiener_max_abs(maxabs_bits), rmax = 1.0f / max_abs;
    const size_t i0 = ((size_t)0 * T + f) * NBINS + b, i1 = ((size_t)1 * T + f) * NBINS + b;
    const size_t j0 = mask_index(0, T, f, b), j1 = mask_index(1, T, f, b);
    const float2 X0 = spec[i0], X1 = spec[i1];
    const float h0 = mix_magnitude(X0), h1 = mix_magnitude(X1);
    float m0[4], m1[4];
    float4 rc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        m0[s] = mags.m[s][j0] * h0; // inference.cpp:175-183: mask x |X|
        m1[s] = mags.m[s][j1] * h1;
        rc[s] = *reinterpret_cast<const float4 *>(R + ((size_t)s * NBINS + b) * 4);
    }
    WienerBin wb;
    wiener_bin_setup(X0, X1, m0, m1, rc, max_abs, rmax, wb);
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        float2 o[2];
        wiener_bin_apply(wb, s, rc[s], max_abs, o);
        y[(((size_t)s * 2 + 0) * T + f) * NBINS + b] = o[0];
        y[(((size_t)s * 2 + 1) * T + f) * NBINS + b] = o[1];
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
