// tools/pk_mfma_probe.hip -- which packed-fp32 VALU forms return wrong results while bf16 MFMA waves share the CU?
//
// Background: with the bf16x3 GEMMs (csrc/gemm_bf16x3.h) co-resident, STFT / iSTFT workgroups
// built with the default SLP vectoriser produced wrong frames; tools/slp_probe.hip reproduces that stand-alone.
// This probe isolates the instruction: a victim kernel evaluates ONE packed form in a long dependent loop next to
// its scalar twin (bitwise-identical by IEEE, pinned with inline asm), a GEMM-like aggressor (ds_write_b128 /
// ds_read_b128 feeding v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 as control) runs on another stream.
// RESULT on MI355X / ROCm 7.2: every form is exact alone and beside the fp32 MFMA; beside the bf16 MFMA the forms
// whose LOW result half selects the HIGH half of src1 (op_sel:[0,1]: "pk_add swap", "pk_mul swap", "pk_add lo<-hi")
// are wrong in ~2 % of all results, everything else (plain, neg, SGPR operand, op_sel_hi, pk_mov, and the
// same-register x + swap(x)) is exact.
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

typedef unsigned long long u64;
__device__ __forceinline__ u64 pack2(float lo, float hi) { return (u64)__float_as_uint(lo) | ((u64)__float_as_uint(hi) << 32); }
__device__ __forceinline__ float lo2(u64 v) { return __uint_as_float((unsigned)v); }
__device__ __forceinline__ float hi2(u64 v) { return __uint_as_float((unsigned)(v >> 32)); }

// The packed forms the SLP-vectorised STFT kernel actually contains (operand modifiers, SGPR operand, pk_mov), each
// next to a scalar twin pinned with inline asm.  OP:
//  0 v_pk_mul_f32            1 v_pk_add_f32              2 v_pk_fma_f32
//  3 v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]   (a - b)
//  4 v_pk_mul_f32 v, v, s[..]                 (SGPR pair operand)
//  5 v_pk_mov_b32 op_sel:[1,0]                (hi of src0, lo of src1)
//  6 v_pk_mul_f32 op_sel_hi:[1,0]             (both halves times src1.lo)
//  7 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (a + swapped b)
template <int OP> __global__ __launch_bounds__(256) void victim(unsigned *bad, int iters)
{
    const float s = 1.0f + 1e-3f * (float)(threadIdx.x & 63);
    u64 p = pack2(s, -s);
    float a = s, b = -s; // scalar twin of (p.lo, p.hi)
    unsigned mism = 0;
    for (int i = 0; i < iters; ++i)
    {
        const float wx = 1.0f + 1e-6f * (float)(i & 1023), wy = 1.0f - 1e-6f * (float)(i & 511);
        const u64 w = pack2(wx, wy);
        if (OP == 0)
        {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else if (OP == 1)
        {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else if (OP == 2)
        {
            const u64 q = pack2(0.75f, 1.25f);
            const float qx = 0.75f, qy = 1.25f;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(w), "v"(q));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(wx), "v"(qx));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(wy), "v"(qy));
        }
        else if (OP == 3)
        {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else if (OP == 4)
        {
            const u64 ws = pack2(1.0f + 1e-6f * (float)(i & 1023), 1.0f - 1e-6f * (float)(i & 511)); // uniform -> SGPRs
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "s"(ws));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else if (OP == 5)
        {
            u64 t;
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(p), "v"(w)); // (p.hi, w.lo)
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(t), "v"(w));              // (p.hi*wx, wx*wy)
            float na, nb;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(na) : "v"(b), "v"(wx));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(nb) : "v"(wx), "v"(wy));
            a = na;
            b = nb;
        }
        else if (OP == 6)
        {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wx));
        }
        else if (OP == 7)
        {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wy));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wx));
        }
        else if (OP == 8) // the LSTM kernel's form: x + swapped x, same register twice, then rescale to stay bounded
        {
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(p));
            float t;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
            a = t;
            b = t;
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else if (OP == 9) // swap on a multiply
        {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wy));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wx));
        }
        else if (OP == 10) // only the low half crosses: lo = a.lo + w.hi, hi = a.hi + w.hi
        {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wy));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wy));
        }
        else // OP 11: only the high half crosses: lo = a.lo + w.lo, hi = a.hi + w.lo
        {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(p), "v"(w));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(wx));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(wx));
        }
        if ((i & 63) == 63)
        {
            mism += (__float_as_uint(lo2(p)) != __float_as_uint(a)) + (__float_as_uint(hi2(p)) != __float_as_uint(b));
            a = s;
            b = -s;
            p = pack2(s, -s);
        }
    }
    if (mism)
        atomicAdd(bad, mism);
}

typedef floatx16 floatx16_;
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

template <int OP> int run(const char *name, hipStream_t sv, hipStream_t sa, unsigned *bad, float *sink)
{
    for (int mode = 0; mode < 3; ++mode) // 0: victim alone, 1: beside fp32 MFMA, 2: beside bf16 MFMA
    {
        CHECK(hipMemset(bad, 0, 4));
        CHECK(hipDeviceSynchronize());
        if (mode == 1)
            hipLaunchKernelGGL(aggressor<false>, dim3(2048), dim3(256), 73728, sa, sink, 20000);
        if (mode == 2)
            hipLaunchKernelGGL(aggressor<true>, dim3(2048), dim3(256), 73728, sa, sink, 40000);
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
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(aggressor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    unsigned *bad;
    float *sink;
    CHECK(hipMalloc(&bad, 4));
    CHECK(hipMalloc(&sink, 4));
    if (run<0>("v_pk_mul_f32", sv, sa, bad, sink) || run<1>("v_pk_add_f32", sv, sa, bad, sink) ||
        run<2>("v_pk_fma_f32", sv, sa, bad, sink) || run<3>("pk_add neg", sv, sa, bad, sink) ||
        run<4>("pk_mul sgpr", sv, sa, bad, sink) || run<5>("pk_mov op_sel", sv, sa, bad, sink) ||
        run<6>("pk_mul op_sel_hi", sv, sa, bad, sink) || run<7>("pk_add swap", sv, sa, bad, sink) ||
        run<8>("pk_add self-swap", sv, sa, bad, sink) || run<9>("pk_mul swap", sv, sa, bad, sink) ||
        run<10>("pk_add lo<-hi", sv, sa, bad, sink) || run<11>("pk_add hi<-lo", sv, sa, bad, sink))
        return 1;
    return 0;
}
