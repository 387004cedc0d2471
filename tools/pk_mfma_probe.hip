// tools/pk_mfma_probe.hip -- do packed-fp32 VALU results change when bf16 MFMA waves share the CU?
//
// Background: while bringing up csrc/gemm_bf16x3.h, STFT / iSTFT workgroups -- whose complex
// arithmetic the compiler had SLP-packed into v_pk_add_f32 / v_pk_mul_f32 / v_pk_mov_b32 -- produced wrong frames
// whenever a GEMM made of v_mfma_f32_32x32x16_bf16 was co-resident; with v_mfma_f32_32x32x2_f32 in the same
// kernel, or with the victims built with -fno-slp-vectorize, everything was bit-exact and deterministic.
// This probe asks the narrow question: a victim kernel evaluates one packed op in a long dependent loop next to
// its scalar twin (bitwise-identical by IEEE), an aggressor kernel issues bf16 (or, as control, fp32) MFMAs on
// another stream; mismatches are counted per op.  RESULT on MI355X / ROCm 7.2: 0 mismatches for v_pk_mul_f32,
// v_pk_add_f32 and v_pk_fma_f32 in every combination -- the packed arithmetic by itself is NOT the mechanism (so
// the LSTM kernel's explicit v_pk_fma_f32 dot is fine, which its pipelined == serial tests confirm); what breaks
// is something else in the SLP-vectorised code shape, still unidentified.  The engine is therefore built with
// -fno-slp-vectorize and guarded by bitwise pipelined-vs-serial tests for both GEMM flavours.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/pk_mfma_probe tools/pk_mfma_probe.hip && tools/pk_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

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

// OP 0: v_pk_mul_f32  1: v_pk_add_f32  2: v_pk_fma_f32.  The packed path is the compiler's own lowering of a
// float2 ext-vector expression; the scalar twin is pinned to v_mul/v_add/v_fma_f32 with inline asm.
template <int OP> __global__ __launch_bounds__(256) void victim(unsigned *bad, int iters)
{
    const float s = 1.0f + 1e-3f * (float)(threadIdx.x & 63);
    float2v p = {s, -s};
    const float2v q = {0.75f, 1.25f};
    float a = s, b = -s; // scalar twin
    unsigned mism = 0;
    for (int i = 0; i < iters; ++i)
    {
        const float wx = 1.0f + 1e-6f * (float)(i & 1023), wy = 1.0f - 1e-6f * (float)(i & 511);
        const float2v w = {wx, wy};
        if (OP == 0)
        {
            p = p * w;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else if (OP == 1)
        {
            p = p + w;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else
        {
            p = __builtin_elementwise_fma(p, w, q);
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(wx), "v"(q.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(wy), "v"(q.y));
        }
        if ((i & 63) == 63)
        {
            const float px = p[0], py = p[1];
            mism += (__float_as_uint(px) != __float_as_uint(a)) + (__float_as_uint(py) != __float_as_uint(b));
            a = s;
            b = -s;
            p[0] = s;
            p[1] = -s;
        }
    }
    if (mism)
        atomicAdd(bad, mism);
}

template <bool BF16> __global__ __launch_bounds__(256) void aggressor(float *sink, int iters)
{
    floatx16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r)
        c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i)
    {
        x[i] = (__bf16)(0.001f * (float)(threadIdx.x + i));
        y[i] = (__bf16)(0.002f * (float)(threadIdx.x ^ i));
    }
    const float fx = 0.001f * (float)threadIdx.x, fy = 0.002f;
    for (int i = 0; i < iters; ++i)
    {
        if (BF16)
        {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c3, 0, 0, 0);
        }
        else
        {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r)
        s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f)
        *sink = s;
}

template <int OP> int run(const char *name, hipStream_t sv, hipStream_t sa, unsigned *bad, float *sink)
{
    for (int mode = 0; mode < 3; ++mode) // 0: victim alone, 1: beside fp32 MFMA, 2: beside bf16 MFMA
    {
        CHECK(hipMemset(bad, 0, 4));
        CHECK(hipDeviceSynchronize());
        if (mode == 1)
            hipLaunchKernelGGL(aggressor<false>, dim3(1024), dim3(256), 0, sa, sink, 400000);
        if (mode == 2)
            hipLaunchKernelGGL(aggressor<true>, dim3(1024), dim3(256), 0, sa, sink, 800000);
        for (int k = 0; k < 8; ++k)
            hipLaunchKernelGGL(victim<OP>, dim3(1024), dim3(256), 0, sv, bad, 200000);
        CHECK(hipStreamSynchronize(sv));
        const bool still = hipStreamQuery(sa) == hipErrorNotReady;
        CHECK(hipDeviceSynchronize());
        unsigned h = 0;
        CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("%-14s %-18s mismatching results: %u%s\n", name,
               mode == 0 ? "alone" : mode == 1 ? "beside f32 MFMA" : "beside bf16 MFMA", h,
               mode && !still ? "  (aggressor finished early)" : "");
    }
    return 0;
}

int main()
{
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    unsigned *bad;
    float *sink;
    CHECK(hipMalloc(&bad, 4));
    CHECK(hipMalloc(&sink, 4));
    if (run<0>("v_pk_mul_f32", sv, sa, bad, sink) || run<1>("v_pk_add_f32", sv, sa, bad, sink) ||
        run<2>("v_pk_fma_f32", sv, sa, bad, sink))
        return 1;
    return 0;
}
