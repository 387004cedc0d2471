// common.h -- shared constants and helpers for the gfx950 UMX engine (device + host).
// Compiled with -ffp-contract=off: elementwise epilogues keep the reference's operation order
// (no silent FMA fusion); dot products / MFMA use explicit fused ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace umx
{

constexpr int NFFT = 4096;  // dsp.hpp:17
constexpr int HOP = 1024;   // dsp.hpp:19
constexpr int NBINS = 2049; // dsp.hpp:49
constexpr int CROP = 1487;  // inference.cpp:55
constexpr int NIN = 2974;   // inference.cpp:41
constexpr int KX = 2976;    // NIN padded to a multiple of 32 (GEMM K tile)
constexpr int NOUT = 4098;  // inference.cpp:53
constexpr int NOUT_PAD = 4352; // fc3's N: two channels of MAGP columns (a multiple of 256, the largest GEMM N tile)
constexpr int MAGP = 2176;     // row pitch of the mask planes [2][T][MAGP] fc3 writes (2049 bins + padding): 17 x 128 floats,
                               // so that a frame's row starts on a 128-byte line and a wave stores whole lines
constexpr int WIENER_BATCH = 200; // wiener.hpp:16
constexpr float WIENER_EPS = 1e-10f;  // wiener.hpp:12
constexpr float WIENER_SCALE = 10.0f; // wiener.hpp:13

constexpr int LSTM_UNITS_PER_WG = 16; // hidden units (x4 gates = 64 gate columns) per workgroup
constexpr int LSTM_THREADS = 512;     // 8 waves: wave w owns k-range [w*Hl/8, (w+1)*Hl/8)
constexpr int LSTM_PERSISTENT_THREADS = 576; // + 1 gate wave in the persistent kernel

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// The streaming kernels (STFT, Wiener statistics / filter + inverse STFT, overlap-add) cover EVERY active track lane of a
// call in one launch: a grid dimension runs over the entries of a LaneSet, and a lane's buffers sit a fixed stride apart
// behind lane 0's (engine.hip allocates them that way).  One launch per kernel and call instead of one per lane: no
// launch gaps and no partly filled last round of workgroups between the lanes (32 lanes: 192 -> 6 launches per call).
constexpr int MAX_TRACK_LANES = 64; // = LSTMB_MAX_TRACKS (lstm_batch.h)
struct LaneSet
{
    int count;
    unsigned char id[MAX_TRACK_LANES]; // track lane of entry i
};

// a / b for a divisor that is reused: q = a * (1/b) corrected once with the exact remainder (Markstein): the
// correctly rounded quotient, i.e. bit-identical to the IEEE division the reference's expression performs
// (inference.cpp:94-95), in 3 instructions instead of the ~11 of v_div_scale / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float div_by(float a, float b, float rcp_b)
{
    const float q = a * rcp_b;
    return fmaf(fmaf(-q, b, a), rcp_b, q);
}

// |X| of one bin (inference.cpp:29, std::abs of a complex float): ONE definition, so that the network input x, the mask x |X|
// products of the Wiener kernels (inference.cpp:175-183) and the debug taps hold the same bits.  Round 5: sqrt(re^2 + im^2)
// (correctly rounded square root of the fp32 sum: within 1.5 ulp of hypotf, which the oracle uses) instead of the device
// library's hypotf (~25 instructions of scaling the spectrogram's range never needs) -- and the SAME value is the divisor of
// the unit phasor below, so a bin needs one square root where it took a hypotf, a sqrtf and two IEEE divisions.
// Valid range: |X| >= ~1e-18 (the squares stay normal fp32 numbers; a 4096-point transform of 16-bit audio has no bin that small
// unless it is exactly zero); below, the squares are subnormal or zero and the value is a magnitude of the right order at best.
__device__ __forceinline__ float mix_magnitude(float2 z) { return sqrtf(z.x * z.x + z.y * z.y); }
// X / |X| (std::polar's phase, wiener.cpp:96-109; arg(0) = 0): the quotients are the correctly rounded ones (div_by: the bits of the
// IEEE divisions this replaced).  |X| below 1e-18 -- where mix_magnitude is no longer accurate and the quotients would not be of unit
// length -- counts as a silent bin (phase 0): a deviation from the reference of the size of the bin itself
// (tests/test_gpu_parity.py::test_wiener_bin_arithmetic_...: a 1e-20 bin)
__device__ __forceinline__ float2 unit_phasor(float2 x)
{
    const float a = mix_magnitude(x);
    const float ra = __builtin_amdgcn_rcpf(a);
    return a > 1e-18f ? make_float2(div_by(x.x, a, ra), div_by(x.y, a, ra)) : make_float2(1.f, 0.f);
}
// Streams: data a kernel reads or writes once and nobody touches again before the kernel has ended (the spectrogram, W_ih x + b_ih, the
// A planes the recurrence leaves for the next GEMM).  Non-temporal accesses keep them from pushing what IS re-used -- weight tiles,
// hand-off granules -- out of the L2s (profiles/r06_ps_store_policy.txt, r06_nt_streams_ab.txt).  -DUMX_NT_STREAMS=0: A/B builds.
#ifndef UMX_NT_STREAMS
#define UMX_NT_STREAMS 1
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load4(const float *p)
{
    if (!UMX_NT_STREAMS)
        return *reinterpret_cast<const float4 *>(p);
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store2(float2 *p, float2 v)
{
    if (!UMX_NT_STREAMS)
        *p = v;
    else
    {
        const nt_f2 t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<nt_f2 *>(p));
    }
}
__device__ __forceinline__ void stream_store4u(uint4 *p, uint4 v)
{
    if (!UMX_NT_STREAMS)
        *p = v;
    else
    {
        const nt_u4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<nt_u4 *>(p));
    }
}
// element (channel c, frame f, bin b) of a mask plane [2][T][MAGP]
__device__ __forceinline__ size_t mask_index(int c, int T, int f, int b) { return ((size_t)c * T + f) * MAGP + b; }

} // namespace umx

#define UMX_HIP_CHECK(expr)                                                                      \
    do                                                                                           \
    {                                                                                            \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
        {                                                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
            return UMX_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)
