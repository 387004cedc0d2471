// tools/slp_probe.hip -- stand-alone reproducer for DESIGN 4.5: the engine's STFT kernel (victim) beside a
// kernel that does nothing but v_mfma_f32_32x32x16_bf16 (aggressor).  Build it twice:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off                    -o tools/slp_probe_slp   tools/slp_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o tools/slp_probe_noslp tools/slp_probe.hip
// Each run computes the reference spectrogram with the victim alone, then repeats it beside the aggressor
// (bf16 MFMA, and fp32 MFMA as control) and counts frames whose bits differ.
#include "../umx.cpp_amd/csrc/stft_kernels.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace umx;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16_ __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                                   \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

// GEMM-like aggressor: every iteration stores operand fragments to LDS (ds_write_b128), barriers, reads them back
// (ds_read_b128) and feeds MFMAs -- the instruction mix of csrc/gemm_bf16x3.h without the global traffic.
template <bool BF16> __global__ __launch_bounds__(256, 2) void aggressor(float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    floatx16_ c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r)
        c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 v = make_uint4(0x3f803f80u + tid, 0x3f003f00u ^ tid, 0x3e803e80u, 0x3e003e00u);
    const int st = (tid >> 1) * 48 + (tid & 1) * 16;
    const int fr = ((wave >> 1) * 64 + (lane & 31)) * 48 + (lane >> 5) * 16;
    for (int i = 0; i < iters; ++i)
    {
        unsigned char *base = lds + (i & 1) * 36864;
#pragma unroll
        for (int p = 0; p < 6; ++p)
            *reinterpret_cast<uint4 *>(base + st + p * 6144) = v;
        __syncthreads();
        bf16x8 a[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
        {
            a[p] = *reinterpret_cast<const bf16x8 *>(base + fr + p * 6144);
            b[p] = *reinterpret_cast<const bf16x8 *>(base + 18432 + fr + p * 6144);
        }
        if (BF16)
        {
#pragma unroll
            for (int p = 0; p < 3; ++p)
            {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p], b[p], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p], b[2 - p], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 - p], b[p], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p], b[p], c3, 0, 0, 0);
            }
        }
        else
        {
#pragma unroll
            for (int p = 0; p < 3; ++p)
            {
                const float fx = (float)a[p][0], fy = (float)b[p][0];
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c3, 0, 0, 0);
            }
        }
        v.x += 0x00010001u;
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r)
        s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f)
        *sink = s;
}

int main()
{
    const int T = 512, N = (T - 1) * HOP, n = N;
    std::vector<float> audio(2 * (size_t)N), win(NFFT);
    for (size_t i = 0; i < audio.size(); ++i)
        audio[i] = 0.3f * sinf(0.001f * (float)i) + 0.1f * sinf(0.0371f * (float)i);
    for (int i = 0; i < NFFT; ++i)
        win[i] = 0.5f * (1.0f - cosf(2.0f * 3.14159265359f * (float)i / (float)NFFT));
    std::vector<float2> t1(256), t2(4096);
    for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 16; ++k)
        {
            const double ph = -2.0 * M_PI * (double)(r * k) / 256.0;
            t1[r * 16 + k] = make_float2((float)cos(ph), (float)sin(ph));
        }
    for (int r = 0; r < 16; ++r)
        for (int j = 0; j < 256; ++j)
        {
            const double ph = -2.0 * M_PI * (double)(r * j) / 4096.0;
            t2[r * 256 + j] = make_float2((float)cos(ph), (float)sin(ph));
        }
    float *d_audio, *d_win, *d_mag, *d_x, *sink;
    float2 *d_t1, *d_t2, *d_spec;
    unsigned *d_max;
    const size_t nspec = (size_t)2 * T * NBINS;
    CHECK(hipMalloc(&d_audio, audio.size() * 4));
    CHECK(hipMalloc(&d_win, NFFT * 4));
    CHECK(hipMalloc(&d_t1, 256 * 8));
    CHECK(hipMalloc(&d_t2, 4096 * 8));
    CHECK(hipMalloc(&d_spec, nspec * 8));
    CHECK(hipMalloc(&d_mag, nspec * 4));
    CHECK(hipMalloc(&d_x, (size_t)T * KX * 4));
    CHECK(hipMalloc(&d_max, 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemcpy(d_audio, audio.data(), audio.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_win, win.data(), NFFT * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_t1, t1.data(), 256 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_t2, t2.data(), 4096 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_max, 0, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    std::vector<float2> ref(nspec), got(nspec);
    hipLaunchKernelGGL(stft_kernel, dim3(T), dim3(256), 0, sv, d_audio, n, N, T, d_win, d_t1, d_t2, d_spec, d_mag, d_x, d_max);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ref.data(), d_spec, nspec * 8, hipMemcpyDeviceToHost));
    for (int mode = 0; mode < 3; ++mode)
    {
        long bad_frames = 0, runs = 0;
        if (mode == 1)
            hipLaunchKernelGGL(aggressor<false>, dim3(2048), dim3(256), 73728, sa, sink, 20000);
        if (mode == 2)
            hipLaunchKernelGGL(aggressor<true>, dim3(2048), dim3(256), 73728, sa, sink, 40000);
        for (int rep = 0; rep < 40; ++rep)
        {
            CHECK(hipMemsetAsync(d_spec, 0, nspec * 8, sv));
            hipLaunchKernelGGL(stft_kernel, dim3(T), dim3(256), 0, sv, d_audio, n, N, T, d_win, d_t1, d_t2, d_spec, d_mag, d_x, d_max);
            CHECK(hipStreamSynchronize(sv));
            CHECK(hipMemcpy(got.data(), d_spec, nspec * 8, hipMemcpyDeviceToHost));
            for (int f = 0; f < T; ++f)
            {
                bool bad = false;
                for (int c = 0; c < 2 && !bad; ++c)
                    bad = memcmp(&got[((size_t)c * T + f) * NBINS], &ref[((size_t)c * T + f) * NBINS], NBINS * 8) != 0;
                bad_frames += bad;
            }
            ++runs;
        }
        const bool still = mode && hipStreamQuery(sa) == hipErrorNotReady;
        CHECK(hipDeviceSynchronize());
        printf("%-18s %ld runs x %d frames: %ld frames differ from the stand-alone result%s\n",
               mode == 0 ? "alone" : mode == 1 ? "beside f32 MFMA" : "beside bf16 MFMA", runs, T, bad_frames,
               mode && !still ? "  (aggressor had finished)" : "");
    }
    return 0;
}
