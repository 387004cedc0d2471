// wiener_istft.h -- Wiener gains + filter (wiener.cpp:270-425) fused with the inverse STFT frame (dsp.cpp:229-258):
// the filtered spectrograms y [4][2][T][2049] complex (339 MB per 60 s segment, written by one kernel and read back by
// the next) never touch HBM.
//
// One 1024-thread workgroup per RUN of consecutive frames of a lane (one workgroup per CU: 158 of its 160 KB of LDS); per frame:
//   phase 1  thread t handles bins b = t + 1024 q (2049 bins, 2-3 per thread): the mixture (2 channels), the four targets' masks
//            (x 2 channels; x |X| = the target magnitudes) and the four R (16 bytes each); forms what does not depend on the output
//            source -- mixture over max_abs, the four PSDs, the inverse of Cxx (wiener.cpp:270-341) -- and then, for each
//            source, y_s = G_s x (wiener.cpp:343-400), written straight into the INPUT of that source's inverse FFT in
//            LDS: one complex 4096-point transform per source carries the left channel in its real and the right channel
//            in its imaginary part (stft_kernels.h), so bin b fills two slots:
//                in[b]        = L[b] + i R[b]                  b <= 2048
//                in[4096 - b] = conj(L[b]) + i conj(R[b])      0 < b < 2048
//   phase 2  the four 256-thread groups transform the four sources side by side (4 x 34 KB of LDS); the last pass stays in registers;
//   phase 3  the reference's per-sample weight (dsp.cpp:248-256) and the overlap-add across the run's frames in registers: a stem
//            sample is stored once (the run's first three hop blocks: wiener_ola_edges_kernel).
// What a frame costs is waiting, not arithmetic: sixteen waves reach every phase together, vector memory returns in order and
// the CU has one address path for it.  Hence: the next frame's first bin is requested under phase 3 (which itself loads nothing behind that
// request: window and the interior hops' sum-square sit in LDS), the second bin and R at the top of phase 1, the second pass's twiddles in
// LDS, the last pass's requested in front of the barrier.  Per frame: 98 KB of spectrogram / magnitudes and 131 KB of R (L2-resident) in,
// 32 KB of stems out.
// WIENER = false is BASELINE config 2: y_s = mag_s * exp(i arg X) (wiener.cpp:96-109 only).
#pragma once
#include "stft_kernels.h"
#include "wiener_kernels.h"

#ifndef WI_PROFILE
#define WI_PROFILE 0 // 1: one workgroup per launch prints where the cycles of a frame go (timing build)
#endif

namespace umx
{

// dynamic LDS of wiener_istft_kernel: the four transforms' buffers + the synthesis window (16 KB) + one hop of the window sum-square (4 KB) + the second pass's twiddles (2 KB): 161,792 of 163,840 B
constexpr size_t WI_LDS_BYTES = (size_t)4 * FFT_LDS_ELEMS * sizeof(float2) + NFFT * sizeof(float) + HOP * sizeof(float) + 256 * sizeof(float2);

// (Two or one source per workgroup -- 2 / 4 workgroups per frame, each repeating phase 1's source-independent part, in exchange for
// more workgroups per CU whose phases overlap -- measured 1.7x / 2.7x slower in round 2; the template parameter is gone.)
// streamed-once inputs: non-temporal loads, so that they do not push the stems' read-modify-write lines out of the L2
typedef float wi_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld_stream(const float2 *p)
{
    const wi_f2 v = __builtin_nontemporal_load(reinterpret_cast<const wi_f2 *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }

typedef unsigned wi_u2 __attribute__((ext_vector_type(2)));
typedef unsigned wi_u4 __attribute__((ext_vector_type(4)));
template <int AUX> __device__ __forceinline__ float2 bld2(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const wi_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <int AUX> __device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX));
}
template <int AUX> __device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const wi_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void bst2(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float2 v)
{
    const wi_u2 t = {__float_as_uint(v.x), __float_as_uint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, soff, 0);
}

template <bool WIENER>
__global__ __launch_bounds__(1024) void wiener_istft_kernel(const float2 *__restrict__ spec, WienerMags mags, int T,
                                                                  const unsigned *__restrict__ maxabs_bits,
                                                                  const float *__restrict__ Rc, const float *__restrict__ window,
                                                                  const float *__restrict__ nw, const float2 *__restrict__ tw1,
                                                                  const float2 *__restrict__ tw2, float2 *__restrict__ frames,
                                                                  float2 *__restrict__ y_dbg, WienerStrides ls, int run_len, OlaOut out)
{
    extern __shared__ __attribute__((aligned(16))) float2 wi_buf[]; // [4 sources][FFT_LDS_ELEMS], then the window [NFFT], then nw of an interior hop block [HOP]
    float *const wi_win = reinterpret_cast<float *>(wi_buf + 4 * FFT_LDS_ELEMS);
    float *const wi_nwp = wi_win + NFFT;
    float2 *const wi_tw1 = reinterpret_cast<float2 *>(wi_nwp + HOP); // tw1: 15 instead of 30 twiddle loads per thread and frame through the CU's one vector-memory path
    constexpr int NSRC = 4, WI_THREADS = 256 * NSRC;
    const LaneSet &lanes = out.lanes;
    {
        const int ln = lanes.id[blockIdx.z]; // grid (runs of run_len frames, 1, lanes): the pointers are lane 0's
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
    constexpr int src0 = 0;
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        wi_win[tid + 1024 * r] = window[tid + 1024 * r]; // (visible after the first frame's barriers)
    // A hop block all four of whose frames exist (3 <= h <= T - 1) has the SAME window sum-square as every other such block: the host adds
    // w^2 of chunks 3, 2, 1, 0 in that order (engine_init.h) -- block 3 stands for them all (in range for every T: the array ends at block T + 2)
    wi_nwp[tid] = nw[3 * HOP + tid];
    if (tid < 256)
        wi_tw1[tid] = tw1[tid];
    // Every streamed access goes through a buffer resource with a scalar base and 32-bit offsets (round 6): a uniform part in an SGPR
    // (frame, channel, source), the thread's part in ONE register per phase -- the 64-bit address arithmetic of a dozen global loads and
    // stores per bin and sample was 9 % of the vector instructions of a kernel that is bound by them.  aux 2 = non-temporal: the inputs
    // are read once (they would push the stems' lines out of the L2).
    const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(spec), 0, (int)((size_t)2 * T * NBINS * 8), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_mag[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        rs_mag[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(mags.m[s]), 0, (int)((size_t)2 * T * MAGP * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Rc), 0, (int)((size_t)4 * NBINS * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_nw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(nw), 0, (int)(((size_t)(T - 1) * HOP + NFFT) * 4), 0x00020000);
    const float max_abs = WIENER ? wiener_max_abs(maxabs_bits) : 1.0f, rmax = 1.0f / max_abs;
    const int f0 = (int)blockIdx.x * run_len, f1 = min(T, f0 + run_len);
    const int g = __builtin_amdgcn_readfirstlane(tid >> 8); // source (of this workgroup's: four waves each); j = tid & 255: thread of its transform
    float2 *const stem = reinterpret_cast<float2 *>(out.p[blockIdx.z][src0 + g]);
    const int n_out = out.n[blockIdx.z];
    const __amdgpu_buffer_rsrc_t rs_stem = __builtin_amdgcn_make_buffer_rsrc(stem, 0, n_out * 8, 0x00020000);
    [[maybe_unused]] long long pf[5] = {0, 0, 0, 0, 0};
    float2 open[3][4]; // the open hop blocks' sums at this thread's positions (see the overlap-add below)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            open[c][q] = make_float2(0.f, 0.f);
    float2 Xq[2][2]; // the mixture and the four masks of the thread's two main bins: 24 registers that ride from one frame's overlap-add to the next frame's gains
    float mq[2][2][4];
#define WI_REQUEST(f_, tl_, q_)                                                                                                        \
    {                                                                                                                                  \
        Xq[q_][0] = bld2<2>(rs_spec, (tl_) * 8, ((0 * T + (f_)) * NBINS + WI_THREADS * (q_)) * 8);                                      \
        Xq[q_][1] = bld2<2>(rs_spec, (tl_) * 8, ((1 * T + (f_)) * NBINS + WI_THREADS * (q_)) * 8);                                      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                                  \
        {                                                                                                                              \
            mq[q_][0][s] = bld1<2>(rs_mag[s], (tl_) * 4, ((0 * T + (f_)) * MAGP + WI_THREADS * (q_)) * 4); /* mask_index(c, T, f, b) */ \
            mq[q_][1][s] = bld1<2>(rs_mag[s], (tl_) * 4, ((1 * T + (f_)) * MAGP + WI_THREADS * (q_)) * 4);                              \
        }                                                                                                                              \
    }
    if (f0 < f1)
    {
        int tl0 = tid;
        asm volatile("" : "+v"(tl0));
        WI_REQUEST(f0, tl0, 0);
    }
    for (int f = f0; f < f1; ++f)
    {
    [[maybe_unused]] long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if (WI_PROFILE)
        c0 = clock64();
    // the thread index is made opaque per frame: everything derived from it (a dozen 64-bit addresses per phase) is formed
    // again where it is used instead of being carried across the phases of every frame (117 spilled registers otherwise)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int j = tl & 255;
    // The streamed inputs (mixture, masks: HBM) of the thread's FIRST bin were requested during the previous frame's overlap-add (WI_REQUEST below;
    // the run's first frame in front of the loop) and those of its second bin are requested here, a bin's arithmetic ahead of their use: no
    // memory latency at the top of a frame with every wave of the workgroup in the same place (rounds 4-5 requested both bins here and waited;
    // both bins during the overlap-add: 24 registers more than that phase has).  (R is L2-resident and stays with its bin: 16 registers.)
    // (R of the first bin goes out in front of that request: it is what the bin waits for next, and vector memory returns in order)
    float4 rc0[4];
    if (WIENER)
    {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            rc0[s] = bld4<0>(rs_rc, tl * 16, (s * NBINS) * 16); // (R is L2-resident and re-read by every frame: default policy)
        __builtin_amdgcn_sched_barrier(0);
    }
    WI_REQUEST(f, tl, 1);
    constexpr int NQ = (NFFT / 2 + WI_THREADS) / WI_THREADS; // 3: bins tl, tl + 1024, and 2048 for thread 0
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
        const int b = tl + WI_THREADS * q;
        if (b > NFFT / 2)
            break;
        float2 X0, X1;
        float m0[4], m1[4];
        if (q < 2)
        {
            X0 = Xq[q < 2 ? q : 0][0];
            X1 = Xq[q < 2 ? q : 0][1];
#pragma unroll
            for (int s = 0; s < 4; ++s)
            {
                m0[s] = mq[q < 2 ? q : 0][0][s];
                m1[s] = mq[q < 2 ? q : 0][1][s];
            }
        }
        else
        {
            X0 = bld2<2>(rs_spec, tl * 8, ((0 * T + f) * NBINS + WI_THREADS * q) * 8);
            X1 = bld2<2>(rs_spec, tl * 8, ((1 * T + f) * NBINS + WI_THREADS * q) * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
            {
                m0[s] = bld1<2>(rs_mag[s], tl * 4, ((0 * T + f) * MAGP + WI_THREADS * q) * 4);
                m1[s] = bld1<2>(rs_mag[s], tl * 4, ((1 * T + f) * MAGP + WI_THREADS * q) * 4);
            }
        }
        const float h0 = mix_magnitude(X0), h1 = mix_magnitude(X1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            m0[s] *= h0; // target magnitude = mask x |X| (inference.cpp:175-183)
            m1[s] *= h1;
        }
        WienerBin wb;
        float4 rc[4];
        float2 p0, p1;
        if (WIENER)
        {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                rc[s] = q == 0 ? rc0[s] : bld4<0>(rs_rc, tl * 16, (s * NBINS + WI_THREADS * q) * 16);
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
                wiener_bin_apply(wb, s, rc[sl], max_abs, o);
            else
            {
                const float ms0 = m0[sl], ms1 = m1[sl];
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
    if (WI_PROFILE)
        c1 = clock64();
    __syncthreads();
    if (WI_PROFILE)
        c2 = clock64();
    float2 *buf = wi_buf + g * FFT_LDS_ELEMS;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        v[r] = buf[fft_pad(j + 256 * r)];
    __syncthreads();
    fft4096<true, true>(v, buf, wi_tw1, tw2, j); // (the result stays in v: sample j + 256 r = v[r])
    if (WI_PROFILE)
        c3 = clock64();
    // ---- the frame's weighted samples: overlap-added on the way out.  Hop block h = samples [h HOP, (h + 1) HOP) of the
    // padded signal is the sum of chunk h - f of the frames f = h - 3 .. h, in ascending f (dsp.cpp:237-257).  A workgroup
    // takes its frames in that order and a position is always the same thread's, so the three OPEN blocks (those that later
    // frames of the run still add to) ride in registers -- open[c][q] = the sum so far of block f + 1 + c at this thread's
    // position q of it -- and every stem sample is stored ONCE, when chunk 0 of the block's last frame has been added (crop of
    // dsp.cpp:203-205: sample p - 2048).  (Round 3 added straight into the stem: 12 device-scope loads and 16 stores per
    // thread and frame, 27.5 GB of memory-side traffic per 32-lane launch against 10.8 GB algorithmic.)  Chunk 3 is a block's
    // first term (0 + c, like the separate kernel).  Only the run's first three blocks also belong to the PREVIOUS run's last
    // frames, whose terms come first: the run's first three frames are kept in `frames` and wiener_ola_edges_kernel adds their
    // chunks to those blocks afterwards, in order, on top of what the previous run flushed at its end (below the loop).
    float2 *dst = frames + ((size_t)(src0 + g) * T + f) * NFFT;
    const int start = f * HOP;
    const bool keep = f - f0 < 3; // one of the run's first three frames
    // The NEXT frame's inputs are requested here, to arrive under this phase and the barrier.  Vector memory returns in order: whatever this phase
    // loaded behind the request would wait for it -- so an INTERIOR frame (hop blocks f .. f + 3 complete: all but the first and last three of
    // a segment) takes window and normalisation from LDS (four values of nw per thread: it repeats with the hop) and requests first; an edge
    // frame reads nw from memory as before and requests last.
    const bool interior = f >= 3 && f + 3 <= T - 1;
    auto overlap_add = [&](auto in_c) {
        constexpr bool IN = decltype(in_c)::value;
        float d4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
        if (IN)
        {
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                d4[q] = wi_nwp[j + 256 * q] + 1e-8f;
                r4[q] = __builtin_amdgcn_rcpf(d4[q]);
            }
            if (f + 1 < f1)
            {
                WI_REQUEST(f + 1, tl, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const int i = j + 256 * r;
            const float2 z = v[r];
            const float w = wi_win[i];
            const float den = IN ? d4[r & 3] : bld1<0>(rs_nw, j * 4, (start + 256 * r) * 4) + 1e-8f;
            // dsp.cpp:248-256: frame * w * 1.0f / 4096 / (nw + 1e-8f), in that order.  The division by 4096 is an exact scaling; the one
            // by den (shared by both channels) as reciprocal + exact-remainder correction (div_by, common.h: the correctly rounded
            // quotient, the bits of the IEEE division istft_frames_kernel performs -- the fused-vs-unfused test compares them):
            // 7 instead of 22 instructions per sample in a kernel that is bound by its VALU instructions (WI_PROFILE)
            const float rden = IN ? r4[r & 3] : __builtin_amdgcn_rcpf(den);
            const float2 val = make_float2(div_by(z.x * w * 1.0f / float(NFFT), den, rden), div_by(z.y * w * 1.0f / float(NFFT), den, rden));
            if (keep)
                dst[i] = val;
            const int c = r >> 2, q = r & 3;
            const float2 a = c < 3 ? open[c < 3 ? c : 0][q] : make_float2(0.f, 0.f);
            const float2 sum = make_float2(a.x + val.x, a.y + val.y);
            if (c == 0)
            {
                // block f is complete
                // (sample start + i - 2048 of the stem: negative or >= n_out = out of the resource's range, the store is dropped)
                if (f >= f0 + 3)
                    bst2(rs_stem, j * 8, (start + 256 * r - NFFT / 2) * 8, sum);
            }
            else
                open[c - 1][q] = sum; // block f + c = block (f + 1) + (c - 1)
        }
        if (!IN && f + 1 < f1)
        {
            WI_REQUEST(f + 1, tl, 0);
        }
    };
    if (interior)
        overlap_add(std::true_type{});
    else
        overlap_add(std::false_type{});
    // the transforms' buffers are free for the next frame's gains: an LDS-only barrier -- __syncthreads() would also wait for the
    // acknowledgements of this frame's stores (vmcnt(0)) with every wave idle, and nothing reads them back
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (WI_PROFILE)
    {
        c4 = clock64();
        pf[0] += c1 - c0;
        pf[1] += c2 - c1;
        pf[2] += c3 - c2;
        pf[3] += c4 - c3;
    }
    }
    // the run's end: blocks f1 .. f1 + 2 hold the terms of this run's last frames; the next run's first frames follow
    // (wiener_ola_edges_kernel), or nothing does (the end of the segment)
    {
        const int j = tid & 255;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
            {
                const int h = f1 + c, s_out = h * HOP + j + 256 * q - NFFT / 2;
                if (h >= f0 + 3 && s_out >= 0 && s_out < n_out)
                    stem[s_out] = open[c][q];
            }
    }
#undef WI_REQUEST
    if (WI_PROFILE && blockIdx.x == 7 && blockIdx.z == 3 && (tid == 0 || tid == 777))
        printf("# wiener_istft thread %d, %d frames: cycles per frame  loads+gains %lld  barrier %lld  transform %lld  weight+overlap-add+drain %lld\n", tid,
               f1 - f0, pf[0] / (f1 - f0), pf[1] / (f1 - f0), pf[2] / (f1 - f0), pf[3] / (f1 - f0));
}

// The first three hop blocks of every run of wiener_istft_kernel: the terms of the run's own frames (kept in `frames`)
// are added to what the previous run's last frames left in the stem, in ascending frame order -- or to zero where no
// earlier frame reaches (the start of the track).  grid (3 blocks x HOP / 256, runs x 4 sources, lanes).
__global__ __launch_bounds__(256) void wiener_ola_edges_kernel(const float2 *__restrict__ frames, size_t frames_stride, int T, int run_len, OlaOut out)
{
    const int run = blockIdx.y >> 2, src = blockIdx.y & 3;
    const int f0 = run * run_len, h = f0 + (int)(blockIdx.x * 256 + threadIdx.x) / HOP, k = (int)(blockIdx.x * 256 + threadIdx.x) % HOP;
    const int s_out = h * HOP + k - NFFT / 2, n_out = out.n[blockIdx.z];
    if (f0 >= T || s_out < 0 || s_out >= n_out)
        return;
    frames += (size_t)out.lanes.id[blockIdx.z] * frames_stride;
    float2 *const stem = reinterpret_cast<float2 *>(out.p[blockIdx.z][src]);
    const int fa = max(0, h - 3), fb = min(T - 1, h);
    // frames before the run left their sum in the stem; f0 == first contributor: nothing was there
    float2 acc = fa < f0 ? stem[s_out] : make_float2(0.f, 0.f);
    for (int f = max(fa, f0); f <= fb; ++f)
    {
        const float2 c = frames[((size_t)src * T + f) * NFFT + (h - f) * HOP + k];
        acc.x += c.x;
        acc.y += c.y;
    }
    stem[s_out] = acc;
}

} // namespace umx
