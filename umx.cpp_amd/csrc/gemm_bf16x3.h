// gemm_bf16x3.h -- the dense stack's GEMMs on the bf16 matrix cores with fp32-class accuracy (the default;
// UMX_CREATE_GEMM_F32 / UMX_GEMM=f32 selects the fp32-MFMA kernel of gemm_kernels.h instead).
//
// Every fp32 operand is split into three bf16 terms, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2) (each subtraction is exact in fp32; the residual after three terms is < 2^-26 |x|),
// and the product a*b is accumulated in fp32 as the six terms of order >= 2^-18:
//     a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1          (dropped: a2 b3, a3 b2, a3 b3  <= 2^-26 |a b|)
// i.e. 6 x v_mfma_f32_32x32x16_bf16 (32 cycles each) in place of 8 x v_mfma_f32_32x32x2_f32 (64 cycles each)
// per 32x32x16 block: 2.7x less matrix-core time at an error below fp32's own rounding of the sum.
//
// Same interface, tile order, and epilogue as gemm_tn_kernel (gemm_kernels.h); what differs:
//   * B (weights) is split once at load time into three bf16 planes [3][N][K] in HBM (1.5x the fp32 bytes);
//   * A (activations) is split while it is staged into LDS (the fc1 input scaling is applied first);
//   * 128 x 128 x 16 tile, LDS holds bf16 planes [plane][row][16 k] with a 48-byte row stride (conflict-free
//     ds_read_b128 / ds_write_b128), double buffered: 2 x 36 KB, two blocks per CU;
//   * lane l of a wave feeds the MFMA row/column l % 32 with the 8 consecutive k of half l / 32.
#pragma once
#include "gemm_common.h"

namespace umx
{

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BX_BK = 16;
constexpr int BX_ROW_BYTES = 48;                      // 16 k x 2 B + 16 B pad
constexpr int BX_PLANE_BYTES = 128 * BX_ROW_BYTES;    // 6,144
constexpr int BX_OPERAND_BYTES = 3 * BX_PLANE_BYTES;  // 18,432
constexpr int BX_BUF_BYTES = 2 * BX_OPERAND_BYTES;    // A planes then B planes: 36,864
constexpr int BX_LDS_BYTES = 2 * BX_BUF_BYTES;        // 73,728

// (x0, x1) -> one dword holding (bf16(x0), bf16(x1)), round-to-nearest-even: a single v_cvt_pk_bf16_f32
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1)
{
    const float2e v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// x -> (x1, x2, x3) for 8 consecutive values, kept as packed dwords all the way: 11 VALU per pair of values
// (element-wise __bf16 code compiled to twice the conversions plus register shuffles)
__device__ __forceinline__ void split3(const float (&x)[8], uint4 &p1, uint4 &p2, uint4 &p3)
{
    unsigned o1[4], o2[4], o3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const unsigned h = cvt_pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        o1[i] = h;
        o2[i] = m;
        o3[i] = cvt_pk_bf16(s0, s1);
    }
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    p3 = make_uint4(o3[0], o3[1], o3[2], o3[3]);
}

// 8 u8 weights -> 8 bf16 values q - 128 (integers in [-128, 127] are exact in bf16: the fp32 value's upper half)
__device__ __forceinline__ uint4 u8x8_to_bf16_centered(unsigned lo, unsigned hi)
{
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const unsigned src = i < 2 ? lo : hi;
        const float f0 = (float)((src >> (16 * (i & 1))) & 255u) - 128.0f, f1 = (float)((src >> (16 * (i & 1) + 8)) & 255u) - 128.0f;
        o[i] = (__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xffff0000u);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// host-side twin of split3 (weights at load time); round-to-nearest-even like v_cvt_pk_bf16_f32
__host__ inline unsigned short bf16_rne_bits(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) // inf / nan: truncate, keep nan quiet
        return (unsigned short)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ inline float bf16_bits_to_float(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ inline void split3_host(float x, unsigned short &p1, unsigned short &p2, unsigned short &p3)
{
    p1 = bf16_rne_bits(x);
    const float r1 = x - bf16_bits_to_float(p1);
    p2 = bf16_rne_bits(r1);
    const float r2 = r1 - bf16_bits_to_float(p2);
    p3 = bf16_rne_bits(r2);
}

// BQ_U8X (u8-resident weights, the default for them): q - 128 is an integer in [-128, 127], EXACT in bf16, so the
// weight needs ONE plane and the product three MFMAs (a1 + a2 + a3).(q - 128) instead of six; the affine map of
// model.cpp:610-616 moves out of the dot product:
//     sum_k a_k (q_k s + o) = s * sum_k a_k (q_k - 128) + (o + 128 s) * sum_k a_k
// The row sums of A come for free from the staging threads (each already holds its 8 values of the K tile).  Half
// the matrix-core time, a third less LDS traffic, no dequantise-and-split VALU work for B.  Against the reference's
// per-weight rounding of q*s+o this differs by that rounding: ~1e-7 of the dot product, one fp32 rounding of the
// sum (tools/gemm_accuracy.py).  UMX_CREATE_U8_DEQUANT selects BQ_U8 (dequantise, then split: bit-identical to
// expanding the weights at load time).
// BQ_F32: GemmTarget::Bq = the three planes [3][N][K] (bf16 bits) split at load time.
// BQ_U8 / BQ_U16 (quantised-resident weights, config 5): GemmTarget::Bq = the file's bytes [N][K]; the staging
// code dequantises (q*scale+offset, model.cpp:610-616) and splits on the fly -- the same three planes bit for
// bit, so both residencies give identical results.
template <int MODE, int BQ> __global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_smem[];
    const GemmTarget tg = args.t[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    int tile_m, tile_n;
    {
        const int gx = args.N / GEMM_BN, gy = args.M / GEMM_BM, total = gx * gy;
        const int chunk = (total + 7) >> 3;
        const int v = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= chunk || v >= total)
            return;
        const int per_group = GEMM_GROUP_M * gx, group = v / per_group, first_m = group * GEMM_GROUP_M;
        const int gsize = min(gy - first_m, GEMM_GROUP_M), in_group = v - group * per_group;
        tile_m = first_m + in_group % gsize;
        tile_n = in_group / gsize;
    }
    const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;
    const int K = args.K, lda = args.lda;

    // staging: lane l of wave w handles row 32 w + (l & 31) and the 8 consecutive k of half l >> 5, of A and of each
    // B plane -- the same (row, half) -> lane pattern the fragment reads use, so the ds_write_b128 are as
    // conflict-free as the ds_read_b128 (row stride 48 B: 16 consecutive rows hit 16 distinct 4-bank groups)
    const int st_row = (tid & 31) + 32 * (tid >> 6), st_half = (tid >> 5) & 1;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tg.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(tg.Bq), 0, 0x7fffffff, 0x00020000);
    constexpr bool U8 = BQ == BQ_U8 || BQ == BQ_U8X;
    constexpr int BEL = U8 ? 1 : 2; // bytes per resident B element (bf16 plane, u16 or u8)
    const int voffA = (st_row * lda + st_half * 8) * 4, voffB = (st_row * K + st_half * 8) * BEL;
    const int soffA0 = m0 * lda * 4, soffB0 = n0 * K * BEL;
    const int plane_stride = args.N * K * 2; // bytes between pre-split B planes
    const float bsc = tg.bs[n0 >= tg.bsplit ? 1 : 0], bof = tg.bo[n0 >= tg.bsplit ? 1 : 0]; // block-uniform
    const int st_lds = st_row * BX_ROW_BYTES + st_half * 16;

    float4 ra0, ra1, rs0, rs1, rm0, rm1;
    uint4 rb1, rb2, rb3;
    rb1 = rb2 = rb3 = make_uint4(0u, 0u, 0u, 0u);
    rs0 = rs1 = rm0 = rm1 = make_float4(0.f, 0.f, 0.f, 0.f);
#define BX_GLOAD(k0)                                                                                   \
    {                                                                                                  \
        ra0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, soffA0 + (k0)*4, 0));      \
        ra1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, soffA0 + (k0)*4 + 16, 0)); \
        if (BQ == BQ_F32)                                                                              \
        {                                                                                              \
            rb1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, soffB0 + (k0)*2, 0));   \
            rb2 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, soffB0 + (k0)*2 + plane_stride, 0));     \
            rb3 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, soffB0 + (k0)*2 + 2 * plane_stride, 0)); \
        }                                                                                              \
        else if (U8)                                                                                   \
        {                                                                                              \
            const uint2 q = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsB, voffB, soffB0 + (k0), 0)); \
            rb1.x = q.x;                                                                               \
            rb1.y = q.y;                                                                               \
        }                                                                                              \
        else                                                                                           \
            rb1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, soffB0 + (k0)*2, 0));   \
        if (MODE == G_FC1)                                                                             \
        {                                                                                              \
            rs0 = *reinterpret_cast<const float4 *>(tg.q0 + (k0) + st_half * 8);                       \
            rs1 = *reinterpret_cast<const float4 *>(tg.q0 + (k0) + st_half * 8 + 4);                   \
            rm0 = *reinterpret_cast<const float4 *>(tg.q1 + (k0) + st_half * 8);                       \
            rm1 = *reinterpret_cast<const float4 *>(tg.q1 + (k0) + st_half * 8 + 4);                   \
        }                                                                                              \
    }
#define BX_SSTORE(buf)                                                                                 \
    {                                                                                                  \
        unsigned char *base = bx_smem + (buf)*BX_BUF_BYTES + st_lds;                                   \
        float4 a0 = ra0, a1 = ra1;                                                                     \
        if (MODE == G_FC1) /* inference.cpp:78-83: x*input_scale + input_mean (F8 order) */            \
        {                                                                                              \
            a0 = scale_shift(a0, rs0, rm0);                                                            \
            a1 = scale_shift(a1, rs1, rm1);                                                            \
        }                                                                                              \
        const float xs[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};                          \
        if (BQ == BQ_U8X) /* this thread's share of the row sum of A, fixed order */                   \
            rowsum += ((xs[0] + xs[1]) + (xs[2] + xs[3])) + ((xs[4] + xs[5]) + (xs[6] + xs[7]));       \
        uint4 p1, p2, p3;                                                                              \
        split3(xs, p1, p2, p3);                                                                        \
        *reinterpret_cast<uint4 *>(base) = p1;                                                         \
        *reinterpret_cast<uint4 *>(base + BX_PLANE_BYTES) = p2;                                        \
        *reinterpret_cast<uint4 *>(base + 2 * BX_PLANE_BYTES) = p3;                                    \
        if (BQ == BQ_U8X)                                                                              \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES) = u8x8_to_bf16_centered(rb1.x, rb1.y); \
        else if (BQ == BQ_F32)                                                                         \
        {                                                                                              \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES) = rb1;                                 \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES + BX_PLANE_BYTES) = rb2;                \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES + 2 * BX_PLANE_BYTES) = rb3;            \
        }                                                                                              \
        else                                                                                           \
        {                                                                                              \
            float ws[8];                                                                               \
            if (BQ == BQ_U8)                                                                           \
            {                                                                                          \
                const float4 lo = deq_u8x4(rb1.x, bsc, bof), hi = deq_u8x4(rb1.y, bsc, bof);           \
                ws[0] = lo.x; ws[1] = lo.y; ws[2] = lo.z; ws[3] = lo.w;                                \
                ws[4] = hi.x; ws[5] = hi.y; ws[6] = hi.z; ws[7] = hi.w;                                \
            }                                                                                          \
            else                                                                                       \
            {                                                                                          \
                const float4 lo = deq_u16x4(make_uint2(rb1.x, rb1.y), bsc, bof);                       \
                const float4 hi = deq_u16x4(make_uint2(rb1.z, rb1.w), bsc, bof);                       \
                ws[0] = lo.x; ws[1] = lo.y; ws[2] = lo.z; ws[3] = lo.w;                                \
                ws[4] = hi.x; ws[5] = hi.y; ws[6] = hi.z; ws[7] = hi.w;                                \
            }                                                                                          \
            uint4 w1, w2, w3;                                                                          \
            split3(ws, w1, w2, w3);                                                                    \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES) = w1;                                  \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES + BX_PLANE_BYTES) = w2;                 \
            *reinterpret_cast<uint4 *>(base + BX_OPERAND_BYTES + 2 * BX_PLANE_BYTES) = w3;             \
        }                                                                                              \
    }

    floatx16 acc00, acc01, acc10, acc11;
    float rowsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        acc00[r] = 0.f;
        acc01[r] = 0.f;
        acc10[r] = 0.f;
        acc11[r] = 0.f;
    }
    // fragment addresses: row (wm*64 + mi*32 + lr), k half lh
    const int fragA = (wm * 64 + lr) * BX_ROW_BYTES + lh * 16;
    const int fragB = BX_OPERAND_BYTES + (wn * 64 + lr) * BX_ROW_BYTES + lh * 16;
#define BX_LD(off) (*reinterpret_cast<const bf16x8 *>(bx_smem + (off)))
#define BX_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
    // smallest terms first, so that each accumulator sees its six products in increasing magnitude
#define BX_TERM(PA, PB)                                                                                \
    {                                                                                                  \
        const bf16x8 a0 = BX_LD(bo + fragA + (PA)*BX_PLANE_BYTES);                                     \
        const bf16x8 a1 = BX_LD(bo + fragA + (PA)*BX_PLANE_BYTES + 32 * BX_ROW_BYTES);                 \
        const bf16x8 b0 = BX_LD(bo + fragB + (PB)*BX_PLANE_BYTES);                                     \
        const bf16x8 b1 = BX_LD(bo + fragB + (PB)*BX_PLANE_BYTES + 32 * BX_ROW_BYTES);                 \
        BX_MFMA(a0, b0, acc00) BX_MFMA(a0, b1, acc01) BX_MFMA(a1, b0, acc10) BX_MFMA(a1, b1, acc11)    \
    }
#define BX_COMPUTE(buf)                                                                                \
    {                                                                                                  \
        const int bo = (buf)*BX_BUF_BYTES;                                                             \
        if (BQ == BQ_U8X)                                                                              \
        {                                                                                              \
            BX_TERM(2, 0) BX_TERM(1, 0) BX_TERM(0, 0)                                                  \
        }                                                                                              \
        else                                                                                           \
        {                                                                                              \
            BX_TERM(2, 0) BX_TERM(0, 2) BX_TERM(1, 1) BX_TERM(1, 0) BX_TERM(0, 1) BX_TERM(0, 0)       \
        }                                                                                              \
    }

    BX_GLOAD(0)
    BX_SSTORE(0)
    __syncthreads();
    const int nk = K / BX_BK;
    for (int kt = 0; kt < nk - 1; ++kt)
    {
        const int cur = kt & 1;
        BX_GLOAD((kt + 1) * BX_BK)
        BX_COMPUTE(cur)
        BX_SSTORE(cur ^ 1)
        __syncthreads();
    }
    BX_COMPUTE((nk - 1) & 1)
    if (BQ == BQ_U8X)
    {
        // acc = sum a (q - 128)  ->  W x = s * acc + (o + 128 s) * rowsum(A).  The two k-halves of every row meet in
        // the LDS buffer the last tile did not use.
        float *rs = reinterpret_cast<float *>(bx_smem + (nk & 1) * BX_BUF_BYTES);
        rs[st_half * 128 + st_row] = rowsum;
        __syncthreads();
        const float o2 = bof + 128.0f * bsc;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                const int ml = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float add = o2 * (rs[ml] + rs[128 + ml]);
                if (mi == 0)
                {
                    acc00[r] = bsc * acc00[r] + add;
                    acc01[r] = bsc * acc01[r] + add;
                }
                else
                {
                    acc10[r] = bsc * acc10[r] + add;
                    acc11[r] = bsc * acc11[r] + add;
                }
            }
    }
#undef BX_GLOAD
#undef BX_SSTORE
#undef BX_LD
#undef BX_MFMA
#undef BX_TERM
#undef BX_COMPUTE
    gemm_epilogue<MODE>(tg, args, m0, n0, wm, wn, lr, lh, acc00, acc01, acc10, acc11);
}

} // namespace umx
