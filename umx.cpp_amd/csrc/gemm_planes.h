// gemm_planes.h -- the dense stack's GEMMs (fc1, W_ih, fc2, fc3: inference.cpp:86,127,143, lstm.cpp:132-135) with BOTH
// operands arriving as bf16 planes, so that the kernel is nothing but LDS-DMA, fragment reads and matrix-core
// instructions (the default flavour; UMX_GEMM=bf16x3 selects gemm_bf16x3.h, UMX_GEMM=f32 gemm_kernels.h).
//
// What gemm_bf16x3.h spent its time on (the matrix pipe busy 41 %, LDS and VALU next to saturated) was
// not the products but the staging: every block re-split its 128 x 16 activation tile into three bf16 terms (44 VALU
// operations and three 16-byte LDS stores per thread and tile, repeated by each of the N/128 blocks that share the
// rows) and dequantised + split its weight tile the same way.  Here
//   * activations are split ONCE, by split_planes_kernel, into planes [3][rows][K] (x = x1 + x2 + x3, bf16 terms,
//     residual < 2^-26 |x|), together with their row sums;
//   * weights are re-encoded ONCE at load time as the integers they are: a u8 weight is q - 128 in ONE bf16 plane
//     (exact), a u16 weight 256 (qh - 128) + (ql - 128) in TWO (both exact), an fp32 weight three split terms; the
//     affine map of model.cpp:610-616 is applied to the accumulated sum with the row sum of A:
//         sum_k a_k (q_k s + o) = s sum_k a_k (q_k - c) + (o + c s) sum_k a_k,   c = 128 or 32896
//   * tiles go global -> LDS by `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write, no VALU), 16 bytes per lane,
//     the XOR swizzle of the LDS layout folded into WHICH 16 bytes a lane fetches;
//   * products per 32x32x16 block: 3 (u8), 5 (u16: a3 x low plane, <= 2^-25 of the leading term, is dropped) or 6
//     (fp32: the bf16x3 rule of gemm_bf16x3.h), fp32 accumulate.
// Block tile (64 WM) x (64 WN) x 32, WM x WN waves, each 2 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16.  With three
// planes per activation the kernel is bound by the bytes it pulls out of the L2s (128 x 128 tiles measured 6.5-10 TB/s
// of L2 -> LDS traffic at 25-40 % of the matrix peak), so the default tile is 256 x 256 (16 waves, one workgroup per CU:
// half the bytes per flop), fed by launches that cover every track lane at once (M = lanes x Tp rows); 128 x 128 remains
// for launches too small to fill the chip with the large tile.  LDS rows are 64 bytes (32 k) as four 16-byte chunks,
// chunk c of row r stored at chunk c ^ ((r >> 2) & 3): a ds_read_b128 group (16 lanes = rows of 4 residues mod 4 x
// 4 values of (r >> 2) & 3) then touches every bank exactly once.  Double-buffered; one barrier per K tile of 24 (u8)
// or 48 MFMAs per wave.  Same XCD-aware tile order and epilogues as gemm_kernels.h.
#pragma once
#include "gemm_bf16x3.h"

namespace umx
{

struct GemmPTarget
{
    const unsigned short *A; // planes [3][a_rows][lda] (bf16 bits); plane p at A + p * a_plane
    const unsigned short *B; // planes [NBP][N][K]; plane p at B + p * N * K
    float *C;
    const float *e0, *e1, *e2, *e3, *q0, *q1, *aux; // as GemmTarget
    float *dbg;
    const float *rs0, *rs1; // row sums of A (rs1 optional: second half of a concatenated A)
    float bs[2], bo2[2];    // NBP < 3: scale, offset + c * scale of the weight tensor(s); rows >= bsplit use [1]
    int bsplit;
};

struct GemmPArgs
{
    GemmPTarget t[4];
    int M, N, K, lda, ldc, T;
    size_t a_plane; // elements between A planes
    int Tp_lane;    // rows per track lane when M spans several lanes (0: one lane); FC3 epilogue, see GemmArgs
    size_t mag_lane, dbg_lane;
};

constexpr int GP_BK = 32;
__host__ __device__ constexpr int gp_lds_bytes(int WM, int WN, int NBP) { return 2 * (3 * 64 * WM + NBP * 64 * WN) * 64; }

// split_planes_kernel: fp32 rows -> three bf16 planes + row sums.  grid (rows_out, 1, targets), 256 threads.
struct SplitArgs
{
    const float *src[4];
    unsigned short *dst[4];
    float *rowsum[4];
    const float *scale[4], *mean[4]; // fc1 prologue x*scale+mean (inference.cpp:78-83, F8 order); nullptr otherwise
    int T, Tp, cols, ld_src, ld_dst, col0_dst; // row m of the grid = frame m % Tp of track lane m / Tp; frames >= T are padding
    size_t plane;                    // elements between output planes
};

__global__ __launch_bounds__(256) void split_planes_kernel(SplitArgs a)
{
    __shared__ float red[4];
    const int m = blockIdx.x, tg = blockIdx.z, tid = threadIdx.x;
    const float *src = a.src[tg] + (size_t)m * a.ld_src;
    unsigned short *dst = a.dst[tg] + (size_t)m * a.ld_dst + a.col0_dst;
    const float *sc = a.scale[tg], *mn = a.mean[tg];
    float sum = 0.f;
    for (int k = tid * 8; k < a.cols; k += 256 * 8)
    {
        float xs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (m % a.Tp < a.T) // rows of the M padding are zero planes
        {
            const float4 v0 = *reinterpret_cast<const float4 *>(src + k), v1 = *reinterpret_cast<const float4 *>(src + k + 4);
            xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w;
            xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
            if (sc)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    xs[j] = xs[j] * sc[k + j] + mn[k + j];
        }
        sum += ((xs[0] + xs[1]) + (xs[2] + xs[3])) + ((xs[4] + xs[5]) + (xs[6] + xs[7]));
        uint4 p1, p2, p3;
        split3(xs, p1, p2, p3);
        *reinterpret_cast<uint4 *>(dst + k) = p1;
        *reinterpret_cast<uint4 *>(dst + a.plane + k) = p2;
        *reinterpret_cast<uint4 *>(dst + 2 * a.plane + k) = p3;
    }
    // row sum in a fixed order: lanes by xor-shuffle, then the four waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        sum += __shfl_xor(sum, off, 64);
    if ((tid & 63) == 0)
        red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0 && a.rowsum[tg])
        a.rowsum[tg][m] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int MODE, int NBP, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16) ? 1 : 2) void gemm_planes_kernel(GemmPArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];
    constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
    constexpr int A_PL = BM * 64, B_PL = BN * 64; // bytes of one plane tile of each operand
    constexpr int BUF_BYTES = 3 * A_PL + NBP * B_PL;
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
    constexpr int A_GROUPS = 3 * (BM / 16), B_GROUPS = NBP * (BN / 16);
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
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lds0 + (buf)*BUF_BYTES + 3 * A_PL + p * B_PL + j * 1024), 16, voffB, \
                                                         (int)(p * b_plane_b + ((long)(n0 + 16 * j) * K + (k0)) * 2), 0, 0); \
            }                                                                                                        \
        }                                                                                                            \
    }

    floatx16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
        acc00[r] = 0.f;
        acc01[r] = 0.f;
        acc10[r] = 0.f;
        acc11[r] = 0.f;
    }
    // fragment of rows (w*64 + mi*32 + lr), k = kk*16 + lh*8 .. +8: logical chunk 2 kk + lh, swizzled by the row
    const int sw = (lr >> 2) & 3; // rows 32 apart share it
    const int fragA = (wm * 64 + lr) * 64, fragB = 3 * A_PL + (wn * 64 + lr) * 64;
#define GP_LD(off) (*reinterpret_cast<const bf16x8 *>(gp_smem + (off)))
#define GP_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
#define GP_TERM(PA, PB, KK)                                                                                          \
    {                                                                                                                \
        const int co = (((KK)*2 + lh) ^ sw) * 16;                                                                    \
        const bf16x8 a0 = GP_LD(bo + fragA + (PA)*A_PL + co);                                                        \
        const bf16x8 a1 = GP_LD(bo + fragA + (PA)*A_PL + 32 * 64 + co);                                              \
        const bf16x8 b0 = GP_LD(bo + fragB + (PB)*B_PL + co);                                                        \
        const bf16x8 b1 = GP_LD(bo + fragB + (PB)*B_PL + 32 * 64 + co);                                              \
        GP_MFMA(a0, b0, acc00) GP_MFMA(a0, b1, acc01) GP_MFMA(a1, b0, acc10) GP_MFMA(a1, b1, acc11)                  \
    }
    // smallest terms first
#define GP_COMPUTE(buf)                                                                                              \
    {                                                                                                                \
        const int bo = (buf)*BUF_BYTES;                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                             \
        {                                                                                                            \
            if (NBP == 1)                                                                                            \
            {                                                                                                        \
                GP_TERM(2, 0, kk) GP_TERM(1, 0, kk) GP_TERM(0, 0, kk)                                                \
            }                                                                                                        \
            else if (NBP == 2) /* B = P_hi + P_lo (both exact): a2 P_lo, a3 P_hi, a1 P_lo, a2 P_hi, a1 P_hi; dropped: */ \
            {                                  /* a3 P_lo <= 2^-17 |a| * 2^7 = 2^-25 of the largest term |a| * 2^15 */ \
                GP_TERM(1, 1, kk) GP_TERM(2, 0, kk) GP_TERM(0, 1, kk) GP_TERM(1, 0, kk) GP_TERM(0, 0, kk)            \
            }                                                                                                        \
            else                                                                                                     \
            {                                                                                                        \
                GP_TERM(2, 0, kk) GP_TERM(0, 2, kk) GP_TERM(1, 1, kk) GP_TERM(1, 0, kk) GP_TERM(0, 1, kk) GP_TERM(0, 0, kk) \
            }                                                                                                        \
        }                                                                                                            \
    }

    GP_DMA(0, 0)
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wave's pieces have landed
    __syncthreads();
    const int nk = K / GP_BK;
    for (int kt = 0; kt < nk - 1; ++kt)
    {
        const int cur = kt & 1;
        GP_DMA(cur ^ 1, (kt + 1) * GP_BK) // the other buffer was last read before the previous barrier
        GP_COMPUTE(cur)
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }
    GP_COMPUTE((nk - 1) & 1)
#undef GP_DMA
#undef GP_LD
#undef GP_MFMA
#undef GP_TERM
#undef GP_COMPUTE
    if (NBP < 3)
    {
        // acc = sum a (q - c)  ->  W x = s * acc + (o + c s) * rowsum(A)
        const int sel = n0 >= tg.bsplit ? 1 : 0;
        const float bsc = tg.bs[sel], o2 = tg.bo2[sel];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float add = o2 * (tg.rs1 ? tg.rs0[m] + tg.rs1[m] : tg.rs0[m]);
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
    GemmTarget et;
    et.C = tg.C;
    et.e0 = tg.e0; et.e1 = tg.e1; et.e2 = tg.e2; et.e3 = tg.e3;
    et.q0 = tg.q0; et.q1 = tg.q1; et.aux = tg.aux; et.dbg = tg.dbg;
    GemmArgs ea;
    ea.ldc = args.ldc;
    ea.T = args.T;
    ea.Tp_lane = args.Tp_lane;
    ea.mag_lane = args.mag_lane;
    ea.dbg_lane = args.dbg_lane;
    gemm_epilogue<MODE>(et, ea, m0, n0, wm, wn, lr, lh, acc00, acc01, acc10, acc11);
}

} // namespace umx
