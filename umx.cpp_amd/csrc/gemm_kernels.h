// gemm_kernels.h -- fp32 MFMA GEMM with fused prologue/epilogue for the dense stack (selected with UMX_GEMM=f32;
// the default flavour is gemm_bf16x3.h, which shares this file's argument structs, tile order and epilogue)
// (inference.cpp:75-99 fc1/bn1/tanh, lstm.cpp:132-135 W_ih x + b_ih for ALL frames at once,
//  inference.cpp:127-140 fc2/bn2/relu, inference.cpp:143-183 fc3/bn3/scale/relu/mask).
//
//   C[M x N] = epilogue( prologue(A[M x K]) * B[N x K]^T )
// A row-major (k contiguous); B is the weight in PyTorch (out, in) row-major layout, i.e. the
// bytes of the ggml file as they are (model.cpp:578-619), zero-padded to N%128==0, K%32==0.
// All four targets run in one launch (blockIdx.z = target).
//
// gfx950 mapping: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, exact fp32 FMA chain;
// 157.3 TFLOP/s peak).  128x128x32 block tile, 4 waves as 2x2, each wave 2x2 MFMA tiles
// (64 accumulator VGPRs).  LDS tiles are [row][k] with a 36-float row stride so that each lane's
// ds_read_b128 of 4 consecutive k is conflict-free; lanes 0-31 take k = kb+{0..3}, lanes 32-63
// take k = kb+{4..7}, and MFMA step s consumes element s of both operands, so A and B agree on
// which k each (half-wave, step) pair means.  Global->LDS goes through registers (prefetch of
// tile k+1 while tile k is multiplied), double-buffered LDS, one barrier per K tile.
// Algorithmic flops per 60 s segment: 453.17 GFLOP.
#pragma once
#include <type_traits>
#include "common.h"

namespace umx
{

typedef float floatx16 __attribute__((ext_vector_type(16)));

enum GemmMode
{
    G_FC1 = 0, // A-prologue x*scale+mean; epilogue bn + tanh
    G_IH = 1,  // epilogue + bias
    G_FC2 = 2, // epilogue bn + relu
    G_FC3 = 3  // epilogue bn, *out_scale + out_mean, relu, * mix_mag -> target magnitude [2][T][2049]
};

struct GemmTarget
{
    const float *A;
    const float *B;
    float *C;
    const float *e0, *e1, *e2, *e3; // bn running_mean, running_var, weight, bias  | IH: e0 = bias
    const float *q0, *q1;           // FC1: input scale, mean [KX]; FC3: output scale, mean [NOUT_PAD]
    const float *aux;               // FC3: mix_mag
    float *dbg;                     // FC3: optional mask tap [T][NOUT]
    // BQ != 0: B stays as stored in the ggml file (u8 / u16, model.cpp:578-619) and is dequantised while it
    // is staged into LDS: w = q * scale + offset in fp32 (model.cpp:610-616).  Rows >= bsplit use the second
    // (scale, offset) pair (W_ih: forward and reverse direction are two tensors).
    const void *Bq;
    float bs[2], bo[2];
    int bsplit;
};

struct GemmArgs
{
    GemmTarget t[4];
    int M, N, K, lda, ldc, T;
    // several track lanes in one launch (gemm_planes.h): M = lanes x Tp_lane rows; the FC3 epilogue's per-lane buffers
    // (mix magnitude in, target magnitude out, mask tap) then sit mag_lane / dbg_lane floats apart.  0 = one lane.
    int Tp_lane, lanes; // lanes: track lanes of the launch (0: M / Tp_lane)
    size_t mag_lane, dbg_lane;
};

// Knob: pin the next tile's global loads at the top of the K tile with sched_barrier (consumers are deferred
// to the LDS store either way).  Measured with flat global loads it helped fc1 only (0.95 -> 0.75 ms) and cost
// 20-30 VGPRs elsewhere; with buffer loads (below) the compiler's own order is best everywhere
// (fc1 0.65 ms), so the default pins nothing.
#ifndef GEMM_PIN_LOADS
#define GEMM_PIN_LOADS(MODE) 0
#endif
#ifndef GEMM_SWIZZLE
#define GEMM_SWIZZLE 1
#endif
#ifndef GEMM_GROUP_M
#define GEMM_GROUP_M 8
#endif
constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32, GEMM_LD = 36;
constexpr int GEMM_LDS_BYTES = 2 * 2 * 128 * GEMM_LD * 4; // 73,728

__device__ __forceinline__ float4 scale_shift(float4 a, float4 sc, float4 mn)
{
    return make_float4(a.x * sc.x + mn.x, a.y * sc.y + mn.y, a.z * sc.z + mn.z, a.w * sc.w + mn.w);
}

// Register budget of the two-slot pipeline (engine.hip): two GEMM blocks (136 VGPRs allocated) must fit a CU
// beside two 8-wave LSTM workgroups of the other slot (104 each): 2 x 104 + 2 x 136 = 480 <= 512 per SIMD lane.
// An LSTM kernel above 120 VGPRs halves the overlapped GEMMs' occupancy (measured: 0.9 -> 2.2 ms).
// tanh for the fc1 epilogue (inference.cpp:99): |x| < 0.5 an odd minimax polynomial through x^11 (9e-8 relative), else
// (1 - e) / (1 + e), e = exp(-2|x|) from v_exp_f32 / v_rcp_f32 (no cancellation there: 1 - e >= 0.63); <= 3e-7 relative
// overall, branch-free, ~18 instructions (the device library's tanhf: ~45 with a divergent branch per element --
// 64 elements per thread at the end of every tile, with the matrix pipe idle).
__device__ __forceinline__ float tanh_epi(float x)
{
    const float ax = fabsf(x), u = x * x;
    const float e = __builtin_amdgcn_exp2f(ax * -2.88539008177792681f); // exp(-2|x|)
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    float q = fmaf(u, -0.006946978159248829f, 0.021472707390785217f);
    q = fmaf(u, q, -0.05393378809094429f);
    q = fmaf(u, q, 0.1333322674036026f);
    q = fmaf(u, q, -0.3333333134651184f);
    const float small = ax * fmaf(u, q, 1.0f);
    return copysignf(ax < 0.5f ? small : big, x);
}

// Epilogue shared by the fp32-MFMA, the bf16x3 and the plane kernels: lane owns column n, rows
// (r&3) + 8*(r>>2) + 4*lh of each 32x32 accumulator tile (the layout of v_mfma_f32_32x32x{2_f32,16_bf16,16_f16}).
// Everything that does not depend on the element is hoisted: the rows of a block belong to ONE track lane (Tp_lane
// is a multiple of the tile height), so the lane index and its offsets are scalars; 32-bit element offsets; groups
// of four rows separated by compiler fences so that the loads of later rows do not pile up in registers (this
// epilogue runs with all 16 waves of a 256 x 256 block at once and nothing to overlap it with).
template <int MODE>
__device__ __forceinline__ void gemm_epilogue(const GemmTarget &tg, const GemmArgs &args, int m0, int n0, int wm, int wn,
                                              int lr, int lh, const floatx16 &acc00, const floatx16 &acc01,
                                              const floatx16 &acc10, const floatx16 &acc11)
{
    // fc3: rows -> (track lane, frame).  Lanes follow each other every Tp_lane rows (>= the tile height), so a block
    // holds rows of at most two lanes: the lane of its first row is a scalar, a row past `m_next` belongs to the next one.
    int f0 = m0;             // frame of row m0 inside its track lane
    unsigned lo = 0, ld = 0; // element offsets of that lane in the magnitude / debug outputs
    const int tpl = args.Tp_lane ? args.Tp_lane : (1 << 30); // rows per lane (one lane: never reached)
    if (MODE == G_FC3 && args.Tp_lane)
    {
        const int ln = __builtin_amdgcn_readfirstlane(m0 / args.Tp_lane);
        f0 = m0 - ln * args.Tp_lane;
        lo = (unsigned)(ln * args.mag_lane);
        ld = (unsigned)(ln * args.dbg_lane);
    }
    const int m_next = tpl - f0; // first row of the block's second lane, relative to m0
    // per lane, once: its first row of the block, that row's frame in either lane, and both as byte offsets
    const int mlb = wm * 64 + 4 * lh;                 // rows of this lane: mlb + (mi*32 + 8 rq + j)
    const int fA = f0 + mlb, fB = mlb - m_next;       // frame of row mlb if it is in the block's first / second lane
    const unsigned dA = ((unsigned)args.mag_lane - (unsigned)args.Tp_lane * NBINS) * 4u; // second lane: + lane stride, - Tp rows
    const unsigned dD = ((unsigned)args.dbg_lane - (unsigned)args.Tp_lane * NOUT) * 4u;
    const bool has_dbg = MODE == G_FC3 && tg.dbg != nullptr;
    // fc3 only (dead code elsewhere): buffer resources of the magnitude output, the mix magnitude and the debug tap
    const int lanes = args.Tp_lane ? (args.lanes ? args.lanes : args.M / args.Tp_lane) : 1;
    const int mag_bytes = MODE == G_FC3 ? (int)((args.Tp_lane ? (size_t)lanes * args.mag_lane : (size_t)2 * args.T * NBINS) * 4) : 0;
    const int dbg_bytes = MODE == G_FC3 && tg.dbg ? (int)((args.Tp_lane ? (size_t)lanes * args.dbg_lane : (size_t)args.T * NOUT) * 4) : 0;
    const __amdgpu_buffer_rsrc_t rs_mag = __builtin_amdgcn_make_buffer_rsrc(tg.C, 0, mag_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_aux = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MODE == G_FC3 ? tg.aux : tg.C), 0, mag_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dbg = __builtin_amdgcn_make_buffer_rsrc(MODE == G_FC3 && tg.dbg ? tg.dbg : tg.C, 0, dbg_bytes, 0x00020000);
    // fc3's debug tap of the mask is a separate (cold) pass over the accumulators, so that the hot pass carries neither
    // its offsets nor a branch per group of rows
    auto pass = [&](auto dbg_tag) {
    constexpr bool DBG = decltype(dbg_tag)::value;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
    {
        const int n = n0 + wn * 64 + ni * 32 + lr;
        float rm = 0.f, sd = 1.f, rsd = 1.f, gw = 1.f, gb = 0.f, osc = 1.f, omn = 0.f;
        if (MODE == G_IH)
            gb = tg.e0[n];
        else
        {
            rm = tg.e0[n];
            sd = sqrtf(tg.e1[n] + 1e-5f); // inference.cpp:94-95
            rsd = 1.0f / sd;
            gw = tg.e2[n];
            gb = tg.e3[n];
        }
        unsigned col = (unsigned)n; // element offset of (row 0 of the block, column n)
        if (MODE == G_FC3)
        {
            osc = tg.q0[n];
            omn = tg.q1[n];
            const int c = n >= NBINS ? 1 : 0;
            col = lo + (unsigned)(c * args.T) * NBINS + (unsigned)(n - c * NBINS);
        }
        // running values of the row loop below (incremented row by row: a handful of constants instead of one literal per
        // row, which the compiler would all keep in SGPRs): byte offsets of (row, column n) in the block's FIRST lane, and
        // the row's frame index relative to the block's SECOND lane (negative while the row is still in the first)
        unsigned run = (col + (unsigned)fA * NBINS) * 4u, drun = (ld + (unsigned)fA * NOUT + (unsigned)n) * 4u;
        int frow = fB;
        // fc3: frames >= T (M padding) and columns >= NOUT (N padding) are dropped by the buffer range check (their
        // byte offset is replaced by one past the end), and so is the whole debug tap when there is none (a resource of
        // zero records): no branch and no 64-bit address arithmetic per element
        const bool col_ok = MODE != G_FC3 || n < NOUT;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
            {
                float ys[4];
                unsigned offs[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int r = 4 * rq + j;
                    const int ml = wm * 64 + mi * 32 + j + 8 * rq + 4 * lh; // row inside the block
                    float y = mi == 0 ? (ni == 0 ? acc00[r] : acc01[r]) : (ni == 0 ? acc10[r] : acc11[r]);
                    if (MODE == G_IH)
                        tg.C[(size_t)(m0 + ml) * args.ldc + n] = y + gb; // lstm.cpp:132-135: W_ih x + b_ih
                    else
                    {
                        y = div_by(y - rm, sd, rsd) * gw + gb; // batchnorm, inference.cpp:93-97 order
                        if (MODE == G_FC1)
                            tg.C[(size_t)(m0 + ml) * args.ldc + n] = tanh_epi(y);
                        else if (MODE == G_FC2)
                            tg.C[(size_t)(m0 + ml) * args.ldc + n] = fmaxf(y, 0.f);
                        else
                        {
                            const bool second = frow >= 0; // the row belongs to the next track lane
                            const int f = second ? frow : frow + tpl;
                            const bool ok = col_ok && f < args.T;
                            ys[j] = fmaxf(y * osc + omn, 0.f); // inference.cpp:161-166
                            // (a lane past the launch's last one lands beyond the resources' range and is dropped)
                            offs[j] = ok ? (DBG ? drun + (second ? dD : 0u) : run + (second ? dA : 0u)) : 0xfffffff0u;
                            run += NBINS * 4;
                            drun += NOUT * 4;
                            frow += 1;
                        }
                    }
                }
                if (MODE == G_FC3 && DBG)
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ys[j]), rs_dbg, offs[j], 0, 0);
                    run += 4 * NBINS * 4;
                    drun += 4 * NOUT * 4;
                    frow += 4;
                }
                else if (MODE == G_FC3)
                {
                    float mix[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        mix[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_aux, offs[j], 0, 0));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ys[j] * mix[j]), rs_mag, offs[j], 0, 0); // inference.cpp:175-183
                    run += 4 * NBINS * 4; // the next group of four rows starts eight rows further
                    drun += 4 * NOUT * 4;
                    frow += 4;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0); // ... and the arithmetic of the next group of rows behind this group's
            }
    }
    };
    if (MODE == G_FC3 && has_dbg)
        pass(std::true_type{});
    pass(std::false_type{});
}

enum GemmBType
{
    BQ_F32 = 0,
    BQ_U8 = 1,
    BQ_U16 = 2,
    BQ_U8X = 3 // bf16x3 kernel only: u8 weights as EXACT bf16 integers, affine map applied to the sum (gemm_bf16x3.h)
};

__device__ __forceinline__ float4 deq_u8x4(unsigned p, float sc, float of)
{
    return make_float4((float)(p & 255u) * sc + of, (float)((p >> 8) & 255u) * sc + of,
                       (float)((p >> 16) & 255u) * sc + of, (float)(p >> 24) * sc + of);
}
__device__ __forceinline__ float4 deq_u16x4(uint2 p, float sc, float of)
{
    return make_float4((float)(p.x & 65535u) * sc + of, (float)(p.x >> 16) * sc + of,
                       (float)(p.y & 65535u) * sc + of, (float)(p.y >> 16) * sc + of);
}

template <int MODE, int BQ> __global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs args)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GemmTarget tg = args.t[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
#if GEMM_SWIZZLE
    // 1-D grid, padded to a multiple of 8.  Workgroups are dealt round-robin to the 8 XCDs (each with its own
    // L2): XCD x gets a contiguous run of tiles, walked in 8-tile-high column groups, so the ~64 tiles an XCD
    // has in flight form a compact patch that shares A and B K-slices through that XCD's L2.
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
#else
    const int m0 = blockIdx.y * GEMM_BM, n0 = blockIdx.x * GEMM_BN;
#endif
    const int K = args.K, lda = args.lda;

    float *const sA0 = smem, *const sB0 = smem + 128 * GEMM_LD;
    constexpr int BUF_STRIDE = 2 * 128 * GEMM_LD;

    // global -> register staging: 4 float4 of A and 4 of B per thread per K tile, as buffer loads: one
    // resource per operand, ONE constant 32-bit per-thread offset (voffset) and a wave-uniform scalar offset
    // (soffset: tile origin + row group + k) -- 2 address VGPRs and no 64-bit vector address arithmetic in the
    // K loop (flat global addressing cost 16-20 VGPRs and pushed the pinned-load variants past the budget).
    const int ld_row = tid >> 3, ld_kc = (tid & 7) * 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tg.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        BQ == BQ_F32 ? (void *)const_cast<float *>(tg.B) : const_cast<void *>(tg.Bq), 0, 0x7fffffff, 0x00020000);
    constexpr int BEL = BQ == BQ_F32 ? 4 : BQ == BQ_U8 ? 1 : 2; // bytes per B element as resident
    const int voffA = (ld_row * lda + ld_kc) * 4, voffB = (ld_row * K + ld_kc) * BEL;
    const int soffA0 = m0 * lda * 4, soffB0 = n0 * K * BEL;
    const float bsc = tg.bs[n0 >= tg.bsplit ? 1 : 0], bof = tg.bo[n0 >= tg.bsplit ? 1 : 0]; // block-uniform
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rsc, rmn;
    uint2 rq0, rq1, rq2, rq3;
    rb0 = rb1 = rb2 = rb3 = rsc = rmn = make_float4(0.f, 0.f, 0.f, 0.f);
    rq0 = rq1 = rq2 = rq3 = make_uint2(0u, 0u);

// Global loads of K tile k0, part j (0..3) = one 32-row group of A and of B.  Loads only: everything that
// consumes the loaded registers (input scaling, dequantisation) happens in UMX_SSTORE, a whole tile of
// MFMAs later, and __builtin_amdgcn_sched_barrier pins the loads where they are written -- left alone, the
// scheduler sinks them to the end of the tile (right before the LDS stores) and every K tile then waits
// out a full memory latency.
#define UMX_AS_F4(v) __builtin_bit_cast(float4, v)
#define UMX_GLOAD_PART(j, ra, rbf, rbq, k0)                                                            \
    {                                                                                                  \
        ra = UMX_AS_F4(__builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, soffA0 + (32 * (j) * lda + (k0)) * 4, 0));  \
        if (BQ == BQ_F32)                                                                              \
            rbf = UMX_AS_F4(__builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, soffB0 + (32 * (j) * K + (k0)) * 4, 0)); \
        else if (BQ == BQ_U8)                                                                          \
            rbq.x = __builtin_amdgcn_raw_buffer_load_b32(rsB, voffB, soffB0 + (32 * (j) * K + (k0)), 0);   \
        else                                                                                           \
            rbq = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsB, voffB, soffB0 + (32 * (j) * K + (k0)) * 2, 0)); \
        if (MODE == G_FC1 && (j) == 0)                                                                 \
        {                                                                                              \
            rsc = *reinterpret_cast<const float4 *>((tg.q0 + (k0)) + (unsigned)ld_kc);                 \
            rmn = *reinterpret_cast<const float4 *>((tg.q1 + (k0)) + (unsigned)ld_kc);                 \
        }                                                                                              \
        if (GEMM_PIN_LOADS(MODE))                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#define UMX_GLOAD(k0)                                                                                  \
    {                                                                                                  \
        UMX_GLOAD_PART(0, ra0, rb0, rq0, k0)                                                           \
        UMX_GLOAD_PART(1, ra1, rb1, rq1, k0)                                                           \
        UMX_GLOAD_PART(2, ra2, rb2, rq2, k0)                                                           \
        UMX_GLOAD_PART(3, ra3, rb3, rq3, k0)                                                           \
    }
#define UMX_FINISH_B(rbf, rbq)                                                                         \
    (BQ == BQ_F32 ? rbf : BQ == BQ_U8 ? deq_u8x4(rbq.x, bsc, bof) : deq_u16x4(rbq, bsc, bof))
// inference.cpp:78-83: x*input_scale + input_mean (F8 order), fused into the A staging
#define UMX_FINISH_A(ra) (MODE == G_FC1 ? scale_shift(ra, rsc, rmn) : ra)
#define UMX_SSTORE(buf)                                                                                \
    {                                                                                                  \
        float *a_ = sA0 + (buf)*BUF_STRIDE + ld_row * GEMM_LD + ld_kc;                                 \
        float *b_ = sB0 + (buf)*BUF_STRIDE + ld_row * GEMM_LD + ld_kc;                                 \
        *reinterpret_cast<float4 *>(a_) = UMX_FINISH_A(ra0);                                           \
        *reinterpret_cast<float4 *>(a_ + 32 * GEMM_LD) = UMX_FINISH_A(ra1);                            \
        *reinterpret_cast<float4 *>(a_ + 64 * GEMM_LD) = UMX_FINISH_A(ra2);                            \
        *reinterpret_cast<float4 *>(a_ + 96 * GEMM_LD) = UMX_FINISH_A(ra3);                            \
        *reinterpret_cast<float4 *>(b_) = UMX_FINISH_B(rb0, rq0);                                      \
        *reinterpret_cast<float4 *>(b_ + 32 * GEMM_LD) = UMX_FINISH_B(rb1, rq1);                       \
        *reinterpret_cast<float4 *>(b_ + 64 * GEMM_LD) = UMX_FINISH_B(rb2, rq2);                       \
        *reinterpret_cast<float4 *>(b_ + 96 * GEMM_LD) = UMX_FINISH_B(rb3, rq3);                       \
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

#define UMX_COMPUTE_G(buf, g)                                                                          \
    {                                                                                                  \
        const float *a = sA0 + (buf)*BUF_STRIDE + (wm * 64 + lr) * GEMM_LD + 4 * lh;                   \
        const float *b = sB0 + (buf)*BUF_STRIDE + (wn * 64 + lr) * GEMM_LD + 4 * lh;                   \
        const float4 a0 = *reinterpret_cast<const float4 *>(a + (g)*8);                                \
        const float4 a1 = *reinterpret_cast<const float4 *>(a + 32 * GEMM_LD + (g)*8);                 \
        const float4 b0 = *reinterpret_cast<const float4 *>(b + (g)*8);                                \
        const float4 b1 = *reinterpret_cast<const float4 *>(b + 32 * GEMM_LD + (g)*8);                 \
        UMX_MFMA4(a0.x, a1.x, b0.x, b1.x)                                                              \
        UMX_MFMA4(a0.y, a1.y, b0.y, b1.y)                                                              \
        UMX_MFMA4(a0.z, a1.z, b0.z, b1.z)                                                              \
        UMX_MFMA4(a0.w, a1.w, b0.w, b1.w)                                                              \
    }
#define UMX_COMPUTE(buf)                                                                               \
    {                                                                                                  \
        UMX_COMPUTE_G(buf, 0) UMX_COMPUTE_G(buf, 1) UMX_COMPUTE_G(buf, 2) UMX_COMPUTE_G(buf, 3)       \
    }
#define UMX_MFMA4(A0, A1, B0, B1)                                                                      \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc00, 0, 0, 0);                              \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc01, 0, 0, 0);                              \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc10, 0, 0, 0);                              \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc11, 0, 0, 0);

    UMX_GLOAD(0)
    UMX_SSTORE(0)
    __syncthreads();
    const int nk = K / GEMM_BK;
    for (int kt = 0; kt < nk - 1; ++kt)
    {
        const int cur = kt & 1;
        UMX_GLOAD((kt + 1) * GEMM_BK)
        UMX_COMPUTE(cur)
        UMX_SSTORE(cur ^ 1)
        __syncthreads();
    }
    UMX_COMPUTE((nk - 1) & 1)
#undef UMX_GLOAD
#undef UMX_SSTORE
#undef UMX_COMPUTE
#undef UMX_COMPUTE_G
#undef UMX_GLOAD_PART
#undef UMX_AS_F4
#undef UMX_FINISH_A
#undef UMX_FINISH_B
#undef UMX_MFMA4

    gemm_epilogue<MODE>(tg, args, m0, n0, wm, wn, lr, lh, acc00, acc01, acc10, acc11);
}

} // namespace umx
