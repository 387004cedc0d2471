// gemm_planes.h -- the dense stack's GEMMs (fc1, W_ih, fc2, fc3: inference.cpp:86,127,143, lstm.cpp:132-135) with BOTH
// operands arriving as fp16 planes, so that the kernel is nothing but LDS-DMA, fragment reads and matrix-core
// instructions (the default flavour of track-batched contexts; UMX_GEMM=bf16x3 selects gemm_bf16x3.h).
//
// What gemm_bf16x3.h spent its time on (the matrix pipe busy 41 %, LDS and VALU next to saturated) was not the
// products but the staging: every block re-split its 128 x 16 activation tile (44 VALU operations and three 16-byte
// LDS stores per thread and tile, repeated by each of the N/128 blocks that share the rows) and dequantised + split
// its weight tile the same way.  Here
//   * activations are split ONCE, by split_planes_kernel, into TWO fp16 planes [2][rows][K]: the row is scaled by a
//     power of two (exact) that brings its largest element into [2^14, 2^15) -- or by the constant 2^14 for tensors
//     bounded by 1 (tanh / LSTM outputs) -- then a1 = fp16(a'), a2 = fp16(a' - a1): a' = a1 + a2 to 2^-22 |a'| worst
//     case (11 + 11 significand bits and the residual's sign), elements more than 2^17 below the row maximum to
//     2^-39 of it; the inverse scale and the fp32 row sum travel with the row;
//   * weights are re-encoded ONCE at load time as the integers they are: a u8 weight is q - 128 in ONE fp16 plane
//     (exact), a u16 weight q - 32896 in TWO whose sum is exact -- P_hi = fp16(q - 32896) and the remainder P_lo, an
//     integer of at most 16 (until the end of round 4: the two bytes, 256 (qh - 128) and ql - 128) --, an fp32 weight two
//     split terms under one power-of-two scale per file tensor; the affine map of model.cpp:610-616 is applied to the
//     accumulated sum with the row sum of A:   sum_k a_k (q_k s + o) = s sum_k a_k (q_k - c) + (o + c s) sum_k a_k,   c = 128 or 32896
//   * tiles go global -> LDS by `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write, no VALU), 16 bytes per lane,
//     the XOR swizzle of the LDS layout folded into WHICH 16 bytes a lane fetches;
//   * products per 32x32x16 block: 2 (u8 weights) or 3 (u16 / fp32 weights: a2 P_hi, a1 P_lo, a1 P_hi; with |P_lo| <=
//     2^-11 |P_hi| the fourth, a2 P_lo, is 2^-22 of the sum -- the size of the activations' own split error -- and is not
//     formed: DESIGN 4.6 (h)), every one exact in the fp32 accumulator's input (22-bit products), fp32 accumulate.
//     (Round 2's first build used three bf16 planes per activation: 3 / 5 / 6 products and 6 bytes per activation
//     element; profiles/r02_v1_*.)
// Block tile (64 WM) x (64 WN) x 32, WM x WN waves, each 2 x 2 MFMA tiles of v_mfma_f32_32x32x16_f16.  The kernel is
// bound by the bytes it pulls out of the L2s as much as by the matrix pipe, so the default tile is 256 x 256 (16 waves,
// one workgroup per CU: half the bytes per flop of 128 x 128), fed by launches that cover every track lane at once
// (M = lanes x Tp rows); 128 x 128 remains for launches too small to fill the chip with the large tile.  LDS rows are
// 64 bytes (32 k) as four 16-byte chunks, chunk c of row r stored at chunk c ^ ((r >> 2) & 3): a ds_read_b128 group
// (16 lanes = rows of 4 residues mod 4 x 4 values of (r >> 2) & 3) then touches every bank exactly once.  STAGES
// buffers (3 where LDS allows), one barrier per K tile.  XCD-aware tile order and epilogues: gemm_common.h.
#pragma once
#include "gemm_bf16x3.h"
#include <cmath>
#include <cstring>

namespace umx
{

struct GemmPTarget
{
    const unsigned short *A; // planes [2][a_rows][lda] (fp16 bits); plane p at A + p * a_plane
    const unsigned short *B; // planes [NBP][N][K]; plane p at B + p * N * K
    float *C;
    const float *e0, *e1, *e2, *e3, *q0, *q1; // as GemmTarget
    const float *rs0, *rs1, *rs2; // row sums of A (rs1, rs2 optional: further parts of a concatenated A -- the skip concat's halves, the two
                                  // directions of a recurrence that wrote its planes itself, lstm_batch.h); added as (rs0 + rs1) + rs2
    const float *rsc;       // per-row inverse scale of A (nullptr: GemmPArgs::a_unscale for every row)
    float bs[2], bo2[2];    // scale, offset + c * scale of the weight tensor(s); columns >= bsplit use [1]
    int bsplit;
};

struct GemmPArgs
{
    GemmPTarget t[4];
    int M, N, K, lda, ldc, T;
    size_t a_plane; // elements between A planes
    float a_unscale; // inverse of the constant scale of A when t[].rsc == nullptr
    int Tp_lane;    // rows per track lane when M spans several lanes (0: one lane); FC3 epilogue, see GemmArgs
    int lanes;      // track lanes of the launch (M >= lanes * Tp_lane, rounded up to the tile)
    size_t mag_lane;
};

constexpr int GP_BK = 32;
constexpr int GP_SPLIT_FIXED_EXP = 14; // tensors bounded by 1 are scaled by 2^14
__host__ __device__ constexpr int gp_stage_bytes(int WM, int WN, int NBP) { return (2 * 64 * WM + NBP * 64 * WN) * 64; }
// three stages where 160 KiB (one workgroup per CU) or 80 KiB (two) allow
__host__ __device__ constexpr int gp_stages(int WM, int WN, int NBP)
{
    return 3 * gp_stage_bytes(WM, WN, NBP) <= (WM * WN >= 16 ? 160 : 80) * 1024 ? 3 : 2;
}
__host__ __device__ constexpr int gp_lds_bytes(int WM, int WN, int NBP) { return gp_stages(WM, WN, NBP) * gp_stage_bytes(WM, WN, NBP); }

// fp32 -> fp16 bits, round to nearest even, subnormals and overflow handled (weights at load time)
__host__ inline unsigned short f16_rne_bits(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u)
        return (unsigned short)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) // rounds to >= 65520: infinity
        return (unsigned short)(sign | 0x7c00u);
    if (u < 0x38800000u) // below 2^-14: subnormal, in units of 2^-24
    {
        if (u < 0x33000000u) // < 2^-25
            return (unsigned short)sign;
        float a;
        memcpy(&a, &u, 4);
        const float scaled = a * 16777216.0f; // exact
        const float r = nearbyintf(scaled);   // default rounding mode: to nearest even
        return (unsigned short)(sign | (unsigned)r);
    }
    const unsigned mant = u & 0x7fffffu, exp = (u >> 23) - 112u; // rebias 127 -> 15
    unsigned h = (exp << 10) | (mant >> 13);
    const unsigned rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
        ++h; // a carry into the exponent is the correct result
    return (unsigned short)(sign | h);
}
__host__ inline float f16_bits_to_float(unsigned short h)
{
    const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    float v;
    if (e == 0)
        v = ldexpf((float)m, -24);
    else if (e == 31)
        v = m ? NAN : INFINITY;
    else
        v = ldexpf((float)(m | 0x400u), (int)e - 25);
    return sign ? -v : v;
}

// split_planes_kernel: fp32 rows -> two fp16 planes of the scaled row + row sum + inverse scale.
// grid (rows_out / 4, 1, targets), 256 threads = four rows; cols <= 512 ITER.
struct SplitArgs
{
    const float *src[4];
    unsigned short *dst[4];
    float *rowsum[4];
    float *rowunscale[4];            // per-row inverse scale out (adaptive scaling); nullptr: the constant 2^GP_SPLIT_FIXED_EXP
    const float *scale[4], *mean[4]; // fc1 prologue x*scale+mean (inference.cpp:78-83, F8 order); nullptr otherwise
    int T, Tp, cols, ld_src, ld_dst, col0_dst; // row m of the grid = frame m % Tp of track lane m / Tp; frames >= T are padding
    int rows_valid;                  // rows >= rows_valid (behind the last lane of the launch) are padding too
    size_t plane;                    // elements between output planes
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split2_f16(const float (&x)[8], float scale, uint4 &p1, uint4 &p2)
{
    f16x8 h1, h2;
#pragma unroll
    for (int j = 0; j < 8; ++j)
    {
        const float v = x[j] * scale; // power of two: exact
        h1[j] = (_Float16)v;          // round to nearest even
        h2[j] = (_Float16)(v - (float)h1[j]);
    }
    p1 = *reinterpret_cast<uint4 *>(&h1);
    p2 = *reinterpret_cast<uint4 *>(&h2);
}

// One WAVE per row (four rows per 256-thread workgroup), ITER x 512 columns: lane l holds the 8 consecutive elements
// k = (it * 64 + l) * 8 of its row, so a wave reads 2 KiB and writes 2 x 1 KiB contiguous per iteration; row sum and
// row maximum by lane exchanges only (round 2's form used one 256-thread workgroup per row -- half of them idle for the
// 1024-column operands -- two barriers and an LDS round trip: 0.72 ms per 32-lane launch at 3.7 TB/s).
template <int ITER> __global__ __launch_bounds__(256) void split_planes_kernel(SplitArgs a)
{
    const int tg = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int m = blockIdx.x * 4 + (tid >> 6); // the grid covers rows_out (a multiple of 256) rows
    const float *src = a.src[tg] + (size_t)m * a.ld_src;
    unsigned short *dst = a.dst[tg] + (size_t)m * a.ld_dst + a.col0_dst;
    const float *sc = a.scale[tg], *mn = a.mean[tg];
    const bool adaptive = a.rowunscale[tg] != nullptr;
    const bool live = m < a.rows_valid && m % a.Tp < a.T; // rows of the M padding are zero planes
    float xs[ITER][8];
    float sum = 0.f, mx = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it)
    {
        const int k = (it * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xs[it][j] = 0.f;
        if (k < a.cols && live)
        {
            const float4 v0 = stream_load4(src + k), v1 = stream_load4(src + k + 4); // (an activation row is split once)
            xs[it][0] = v0.x; xs[it][1] = v0.y; xs[it][2] = v0.z; xs[it][3] = v0.w;
            xs[it][4] = v1.x; xs[it][5] = v1.y; xs[it][6] = v1.z; xs[it][7] = v1.w;
            if (sc)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    xs[it][j] = xs[it][j] * sc[k + j] + mn[k + j];
        }
        sum += ((xs[it][0] + xs[it][1]) + (xs[it][2] + xs[it][3])) + ((xs[it][4] + xs[it][5]) + (xs[it][6] + xs[it][7]));
#pragma unroll
        for (int j = 0; j < 8; ++j)
            mx = fmaxf(mx, fabsf(xs[it][j]));
    }
    // row sum in a fixed order (per lane in column order, then the lanes by xor-exchange); row maximum alongside
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        sum += __shfl_xor(sum, off, 64);
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    float scale = (float)(1 << GP_SPLIT_FIXED_EXP), unscale = 1.0f / (float)(1 << GP_SPLIT_FIXED_EXP);
    if (adaptive)
    {
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f)
        {
            int x;
            (void)frexpf(mx, &x); // mx = f 2^x, f in [0.5, 1)
            e = min(max(GP_SPLIT_FIXED_EXP + 1 - x, -100), 100);
        }
        scale = ldexpf(1.0f, e);
        unscale = ldexpf(1.0f, -e);
    }
    if (lane == 0)
    {
        if (a.rowsum[tg])
            a.rowsum[tg][m] = sum;
        if (adaptive)
            a.rowunscale[tg][m] = unscale;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it)
    {
        const int k = (it * 64 + lane) * 8;
        if (k < a.cols)
        {
            uint4 p1, p2;
            split2_f16(xs[it], scale, p1, p2);
            stream_store4u(reinterpret_cast<uint4 *>(dst + k), p1); // (gigabytes per launch: gone from the L2s long before the GEMM asks)
            stream_store4u(reinterpret_cast<uint4 *>(dst + a.plane + k), p2);
        }
    }
}

// The same for operands that ALL targets derive from ONE source row (fc1's input: x * input_scale + input_mean per target,
// inference.cpp:78-83): the row is read once and the targets' planes are written one after the other -- round 3 launched the
// kernel above per target and read the 1.07 GB of x four times per 32-lane step (7.7 GB of counter traffic, 1.52 ms).
// grid (rows_out / 4), 256 threads = four rows; same arithmetic per target, same bits.
template <int ITER> __global__ __launch_bounds__(256) void split_planes_shared_kernel(SplitArgs a, int ntargets)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int m = blockIdx.x * 4 + (tid >> 6);
    const float *src = a.src[0] + (size_t)m * a.ld_src;
    const bool live = m < a.rows_valid && m % a.Tp < a.T;
    float x[ITER][8];
#pragma unroll
    for (int it = 0; it < ITER; ++it)
    {
        const int k = (it * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            x[it][j] = 0.f;
        if (k < a.cols && live)
        {
            const float4 v0 = stream_load4(src + k), v1 = stream_load4(src + k + 4); // (an activation row is split once)
            x[it][0] = v0.x; x[it][1] = v0.y; x[it][2] = v0.z; x[it][3] = v0.w;
            x[it][4] = v1.x; x[it][5] = v1.y; x[it][6] = v1.z; x[it][7] = v1.w;
        }
    }
    for (int tg = 0; tg < ntargets; ++tg)
    {
        unsigned short *dst = a.dst[tg] + (size_t)m * a.ld_dst + a.col0_dst;
        const float *sc = a.scale[tg], *mn = a.mean[tg];
        float xs[ITER][8];
        float sum = 0.f, mx = 0.f;
#pragma unroll
        for (int it = 0; it < ITER; ++it)
        {
            const int k = (it * 64 + lane) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xs[it][j] = 0.f;
            if (k < a.cols && live)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    xs[it][j] = x[it][j] * sc[k + j] + mn[k + j];
            sum += ((xs[it][0] + xs[it][1]) + (xs[it][2] + xs[it][3])) + ((xs[it][4] + xs[it][5]) + (xs[it][6] + xs[it][7]));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                mx = fmaxf(mx, fabsf(xs[it][j]));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            sum += __shfl_xor(sum, off, 64);
            mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        }
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f)
        {
            int xe;
            (void)frexpf(mx, &xe);
            e = min(max(GP_SPLIT_FIXED_EXP + 1 - xe, -100), 100);
        }
        const float scale = ldexpf(1.0f, e), unscale = ldexpf(1.0f, -e);
        if (lane == 0)
        {
            a.rowsum[tg][m] = sum;
            a.rowunscale[tg][m] = unscale;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it)
        {
            const int k = (it * 64 + lane) * 8;
            if (k < a.cols)
            {
                uint4 p1, p2;
                split2_f16(xs[it], scale, p1, p2);
                stream_store4u(reinterpret_cast<uint4 *>(dst + k), p1);
                stream_store4u(reinterpret_cast<uint4 *>(dst + a.plane + k), p2);
            }
        }
    }
}

// Wave tile 64 x 64 (MI = 2 matrix tiles of 32 rows along M): sixteen waves per 256 x 256 block, all in lock step on one
// barrier per K tile.  The eight-wave 128 x 64 form of round 3 lives on as gemm_planes_pp.h, where the two waves of a SIMD take
// turns at the matrix pipe: the default for every 256 x 256 launch (UMX_GEMM_PP=0 selects this kernel); it gives the bits of this
// kernel, which stays the only form for the 128 x 128 tiles of small launches.
template <int MODE, int NBP, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16) ? 1 : 2) void gemm_planes_kernel(GemmPArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];
    constexpr int MI = 2;
    constexpr int BM = 32 * MI * WM, BN = 64 * WN, NW = WM * WN;
    constexpr int A_PL = BM * 64, B_PL = BN * 64; // bytes of one plane tile of each operand
    constexpr int BUF_BYTES = 2 * A_PL + NBP * B_PL, STAGES = gp_stages(BM / 64, WN, NBP);
    static_assert(NBP == 1 || NBP == 2, "weight planes: 1 (u8) or 2 (u16, fp32)");
    const GemmPTarget tg = args.t[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, lr = lane & 31, lh = lane >> 5;
    int tile_m, tile_n;
    {
        const int gx = args.N / BN, gy = args.M / BM, total = gx * gy;
        const int chunk = (total + 7) >> 3;
        const int v = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= chunk || v >= total)
            return;
        const int per_group = GEMM_GROUP_M * gx, group = v / per_group, first_m = group * GEMM_GROUP_M;
        const int gsize = min(gy - first_m, GEMM_GROUP_M), in_group = v - group * per_group;
        tile_m = first_m + in_group % gsize;
        tile_n = in_group / gsize;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = args.K, lda = args.lda;

    // ---- LDS-DMA staging.  One wave-instruction fills 1 KiB = 16 rows x 4 chunks of a plane tile; lane L lands at
    // row 16 j + L/4, physical chunk L%4, and therefore FETCHES logical chunk (L%4) ^ ((L/16) & 3) of that row.
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(tg.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(tg.B), 0, 0x7fffffff, 0x00020000);
    const int st_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int voffA = ((lane >> 2) * lda) * 2 + st_chunk * 16, voffB = ((lane >> 2) * K) * 2 + st_chunk * 16;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)gp_smem;
    const long a_plane_b = (long)args.a_plane * 2, b_plane_b = (long)args.N * K * 2;
    // the 16-row groups of all plane tiles are dealt round-robin to the waves
    constexpr int A_GROUPS = 2 * (BM / 16), B_GROUPS = NBP * (BN / 16), DMA_PER_WAVE = (A_GROUPS + B_GROUPS) / NW;
    static_assert(A_GROUPS % NW == 0 && B_GROUPS % NW == 0 && DMA_PER_WAVE < 16, "vmcnt bookkeeping below");
#define GP_DMA(buf, k0)                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int i0 = 0; i0 < A_GROUPS; i0 += NW)                                                  \
        {                                                                                                            \
            const int i = i0 + wave;                                                                                 \
            if (A_GROUPS % NW == 0 || i < A_GROUPS)                                                                  \
            {                                                                                                        \
                const int p = i / (BM / 16), j = i % (BM / 16);                                                      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(size_t)(lds0 + (buf)*BUF_BYTES + p * A_PL + j * 1024), 16, voffA, \
                                                         (int)(p * a_plane_b + ((long)(m0 + 16 * j) * lda + (k0)) * 2), 0, 0); \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int i0 = 0; i0 < B_GROUPS; i0 += NW)                                                  \
        {                                                                                                            \
            const int i = i0 + wave;                                                                                 \
            if (B_GROUPS % NW == 0 || i < B_GROUPS)                                                                  \
            {                                                                                                        \
                const int p = i / (BN / 16), j = i % (BN / 16);                                                      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lds0 + (buf)*BUF_BYTES + 2 * A_PL + p * B_PL + j * 1024), 16, voffB, \
                                                         (int)(p * b_plane_b + ((long)(n0 + 16 * j) * K + (k0)) * 2), 0, 0); \
            }                                                                                                        \
        }                                                                                                            \
    }

    floatx16 acc[MI][2]; // [32-row tile along M][32-column tile along N] of the wave's (32 MI) x 64 block
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            acc[mi][0][r] = 0.f;
            acc[mi][1][r] = 0.f;
        }
    // fragment of rows (w*32*MI + mi*32 + lr), k = kk*16 + lh*8 .. +8: logical chunk 2 kk + lh, swizzled by the row
    const int sw = (lr >> 2) & 3; // rows 32 apart share it
    const int fragA = (wm * 32 * MI + lr) * 64, fragB = 2 * A_PL + (wn * 64 + lr) * 64;
#define GP_LD(off) (*reinterpret_cast<const f16x8 *>(gp_smem + (off)))
#define GP_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
#define GP_TERM(PA, PB, KK)                                                                                          \
    {                                                                                                                \
        const int co = (((KK)*2 + lh) ^ sw) * 16;                                                                    \
        const f16x8 b0 = GP_LD(bo + fragB + (PB)*B_PL + co);                                                         \
        const f16x8 b1 = GP_LD(bo + fragB + (PB)*B_PL + 32 * 64 + co);                                               \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                            \
        {                                                                                                            \
            const f16x8 am = GP_LD(bo + fragA + (PA)*A_PL + mi * 32 * 64 + co);                                      \
            GP_MFMA(am, b0, acc[mi][0]) GP_MFMA(am, b1, acc[mi][1])                                                  \
        }                                                                                                            \
    }
    // smallest terms first
#define GP_COMPUTE(buf)                                                                                              \
    {                                                                                                                \
        const int bo = (buf)*BUF_BYTES;                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
        {                                                                                                            \
            if (NBP == 1)                                                                                            \
            {                                                                                                        \
                GP_TERM(1, 0, kk) GP_TERM(0, 0, kk)                                                                  \
            }                                                                                                        \
            else /* B = P_hi + P_lo, |P_lo| <= 2^-11 |P_hi|: a2 P_hi, a1 P_lo, a1 P_hi (a2 P_lo, 2^-22 of the sum, is not formed) */ \
            {                                                                                                        \
                GP_TERM(1, 0, kk) GP_TERM(0, 1, kk) GP_TERM(0, 0, kk)                                                \
            }                                                                                                        \
        }                                                                                                            \
    }

    // s_waitcnt vmcnt(n): all but the n most recent DMA instructions of this wave have landed
#define GP_WAIT(n) __builtin_amdgcn_s_waitcnt(0x0f70 | (n))
    const int nk = K / GP_BK;
    GP_DMA(0, 0)
    if (STAGES == 3 && nk > 1)
    {
        GP_DMA(1, GP_BK)
        GP_WAIT(DMA_PER_WAVE);
    }
    else
        GP_WAIT(0);
    __syncthreads();
// timing experiments (results are garbage): which side of the main loop sets its pace?
#ifndef GP_EXPERIMENT_NO_DMA
#define GP_EXPERIMENT_NO_DMA 0
#endif
#ifndef GP_EXPERIMENT_NO_MFMA
#define GP_EXPERIMENT_NO_MFMA 0
#endif
#ifndef GP_PROFILE
#define GP_PROFILE 0 // 1: one wave of one workgroup per launch prints where the cycles of its main loop go (timing build)
#endif
    [[maybe_unused]] long long pf[4] = {0, 0, 0, 0};
    if (STAGES == 3)
    {
        int cur = 0; // stage of tile kt; kt + 2 goes where kt - 1 was (last read before the previous barrier)
        for (int kt = 0; kt < nk - 2; ++kt)
        {
            const int nxt = cur == 0 ? 2 : cur - 1; // (cur + 2) % 3
            [[maybe_unused]] long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (GP_PROFILE)
                c0 = clock64();
            if (!GP_EXPERIMENT_NO_DMA)
                GP_DMA(nxt, (kt + 2) * GP_BK)
            if (GP_PROFILE)
                c1 = clock64();
            if (!GP_EXPERIMENT_NO_MFMA)
                GP_COMPUTE(cur)
            if (GP_PROFILE)
                c2 = clock64();
            GP_WAIT(DMA_PER_WAVE); // tile kt + 1 has landed, kt + 2 may be in flight
            if (GP_PROFILE)
                c3 = clock64();
            __syncthreads();
            if (GP_PROFILE)
            {
                const long long c4 = clock64();
                pf[0] += c1 - c0;
                pf[1] += c2 - c1;
                pf[2] += c3 - c2;
                pf[3] += c4 - c3;
            }
            cur = cur == 2 ? 0 : cur + 1;
        }
        if (GP_PROFILE && blockIdx.x == 64 && blockIdx.z == 0 && (tid == 0 || tid == 64 * 7))
            printf("# gemm_planes<%d,%d> wave %d trips %d: cycles per trip  dma-issue %lld  fragments+mfma %lld  wait-dma %lld  barrier %lld\n", MODE, NBP,
                   wave, nk - 2, pf[0] / (nk - 2), pf[1] / (nk - 2), pf[2] / (nk - 2), pf[3] / (nk - 2));
        if (nk > 1)
        {
            GP_COMPUTE(cur)
            GP_WAIT(0);
            __syncthreads();
            cur = cur == 2 ? 0 : cur + 1;
        }
        GP_COMPUTE(cur)
    }
    else
    {
        for (int kt = 0; kt < nk - 1; ++kt)
        {
            const int cur = kt & 1;
            if (!GP_EXPERIMENT_NO_DMA)
                GP_DMA(cur ^ 1, (kt + 1) * GP_BK) // the other buffer was last read before the previous barrier
            if (!GP_EXPERIMENT_NO_MFMA)
                GP_COMPUTE(cur)
            GP_WAIT(0);
            __syncthreads();
        }
        GP_COMPUTE((nk - 1) & 1)
    }
#undef GP_WAIT
#undef GP_DMA
#undef GP_LD
#undef GP_MFMA
#undef GP_TERM
#undef GP_COMPUTE
    {
        // acc = 2^e_m sum a (q - c)  ->  W x = (s 2^-e_m) acc + (o + c s) rowsum(A).  The two per-row factors are
        // computed once per row of the block and handed to the lanes through LDS (the stage buffers are free now);
        // read per group of four rows, so that they never crowd the accumulators out of the register file.
        const int sel = n0 >= tg.bsplit ? 1 : 0;
        const float bsc = tg.bs[sel], o2 = tg.bo2[sel];
        float *const fx = reinterpret_cast<float *>(gp_smem); // [2][BM]
        __syncthreads();                                      // every wave is done with the last stage
        for (int i = tid; i < BM; i += 64 * NW)
        {
            const int m = m0 + i;
            fx[i] = bsc * (tg.rsc ? tg.rsc[m] : args.a_unscale); // a power of two times s: exact scaling
            fx[BM + i] = o2 * (tg.rs2 ? (tg.rs0[m] + tg.rs1[m]) + tg.rs2[m] : tg.rs1 ? tg.rs0[m] + tg.rs1[m] : tg.rs0[m]);
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
            {
                const int ml = wm * 32 * MI + mi * 32 + 8 * rq + 4 * lh; // rows (r & 3) + 8 (r >> 2) + 4 lh, r = 4 rq + j
                const float4 mu = *reinterpret_cast<const float4 *>(fx + ml), ad = *reinterpret_cast<const float4 *>(fx + BM + ml);
                const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, ads[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int r = 4 * rq + j;
                    acc[mi][0][r] = mus[j] * acc[mi][0][r] + ads[j];
                    acc[mi][1][r] = mus[j] * acc[mi][1][r] + ads[j];
                }
                asm volatile("" ::: "memory"); // keep the next group's reads behind this group's arithmetic
            }
    }
    GemmTarget et;
    et.C = tg.C;
    et.e0 = tg.e0; et.e1 = tg.e1; et.e2 = tg.e2; et.e3 = tg.e3;
    et.q0 = tg.q0; et.q1 = tg.q1;
    GemmArgs ea;
    ea.M = args.M;
    ea.ldc = args.ldc;
    ea.T = args.T;
    ea.Tp_lane = args.Tp_lane;
    ea.lanes = args.lanes;
    ea.mag_lane = args.mag_lane;
    // the shared epilogue takes a 64 x 64 wave tile at rows m0 + 64 wm: a 128 x 64 wave tile is two of them
#pragma unroll
    for (int half = 0; half < MI / 2; ++half)
        gemm_epilogue<MODE>(et, ea, m0 + wm * 32 * MI + half * 64, n0, 0, wn, lr, lh, acc[2 * half][0], acc[2 * half][1], acc[2 * half + 1][0],
                            acc[2 * half + 1][1]);
}

} // namespace umx
