// gemm_common.h -- what the dense stack's GEMM flavours share: argument structs, tile order knobs, the fused epilogues
// (inference.cpp:75-99 fc1/bn1/tanh, lstm.cpp:132-135 W_ih x + b_ih for ALL frames at once,
//  inference.cpp:127-140 fc2/bn2/relu, inference.cpp:143-174 fc3/bn3/scale/relu = the MASK).
//
//   C[M x N] = epilogue( prologue(A[M x K]) * B[N x K]^T )
// A row-major (k contiguous); B is the weight in PyTorch (out, in) row-major layout, zero-padded to N % 128 == 0,
// K % 32 == 0.  All four targets run in one launch (blockIdx.z = target).  The kernels: gemm_planes.h (track-batched
// contexts) and gemm_bf16x3.h (single-track contexts).  (Rounds 1-2 also carried an fp32-MFMA kernel, 157 TFLOP/s peak:
// slower and, against float64, less accurate than both split-operand flavours -- profiles/r02_accuracy_vs_float64.txt --
// removed in round 3.)
//
// fc3 writes the MASK, not the target magnitude: out[c][t][b] at pitch MAGP (common.h) -- whole 128-byte lines per store
// instruction -- and the consumers (Wiener statistics / filter, mixture-phase estimate) form mask x |X| themselves
// (inference.cpp:175-183: one fp32 multiply, the same bits wherever it is done).  Round 2's epilogue read |X| back
// (5.4 GB per 32-lane launch) and stored rows of 2049 floats that straddled lines: 36.9 GB of HBM-side traffic per
// launch against 12.3 GB algorithmic (profiles/r02_v4_pmc_fetch_write_per_kernel.csv).
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
    G_FC3 = 3  // epilogue bn, *out_scale + out_mean, relu -> mask [2][T][MAGP] (columns: channel c at c * MAGP)
};

struct GemmTarget
{
    const float *A;
    const float *B;
    float *C;
    const float *e0, *e1, *e2, *e3; // bn running_mean, running_var, weight, bias  | IH: e0 = bias
    const float *q0, *q1;           // FC1: input scale, mean [KX]; FC3: output scale, mean [NOUT_PAD]
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
    // several track lanes in one launch (gemm_planes.h): M = lanes x Tp_lane rows; the FC3 epilogue's per-lane mask
    // buffers then sit mag_lane floats apart.  0 = one lane.
    int Tp_lane, lanes; // lanes: track lanes of the launch (0: M / Tp_lane)
    size_t mag_lane;
};

// Knob: pin the next tile's global loads at the top of the K tile with sched_barrier (consumers are deferred
// to the LDS store either way).  Measured with flat global loads it helped fc1 only (0.95 -> 0.75 ms) and cost
// 20-30 VGPRs elsewhere; with buffer loads (below) the compiler's own order is best everywhere
// (fc1 0.65 ms), so the default pins nothing.
#ifndef GEMM_GROUP_M
#define GEMM_GROUP_M 8
#endif
constexpr int GEMM_BM = 128, GEMM_BN = 128;

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

// Epilogue shared by the bf16x3 and the plane kernels: lane owns column n, rows
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
    int f0 = m0;  // frame of row m0 inside its track lane
    unsigned lo = 0; // element offset of that lane in the mask output
    const int tpl = args.Tp_lane ? args.Tp_lane : (1 << 30); // rows per lane (one lane: never reached)
    if (MODE == G_FC3 && args.Tp_lane)
    {
        const int ln = __builtin_amdgcn_readfirstlane(m0 / args.Tp_lane);
        f0 = m0 - ln * args.Tp_lane;
        lo = (unsigned)(ln * args.mag_lane);
    }
    const int m_next = tpl - f0; // first row of the block's second lane, relative to m0
    // per lane, once: its first row of the block, that row's frame in either lane, and both as byte offsets
    const int mlb = wm * 64 + 4 * lh;                 // rows of this lane: mlb + (mi*32 + 8 rq + j)
    const int fA = f0 + mlb, fB = mlb - m_next;       // frame of row mlb if it is in the block's first / second lane
    const unsigned dA = ((unsigned)args.mag_lane - (unsigned)args.Tp_lane * MAGP) * 4u; // second lane: + lane stride, - Tp rows
    // fc3 only (dead code elsewhere): buffer resource of the mask output
    const int lanes = args.Tp_lane ? (args.lanes ? args.lanes : args.M / args.Tp_lane) : 1;
    // (unsigned: 48 lanes x T = 2584 is 2.16e9 bytes; engine.hip refuses launches beyond 2^32)
    const unsigned mag_bytes = MODE == G_FC3 ? (unsigned)((args.Tp_lane ? (size_t)lanes * args.mag_lane : (size_t)2 * args.T * MAGP) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t rs_mag = __builtin_amdgcn_make_buffer_rsrc(tg.C, 0, mag_bytes, 0x00020000);
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
        bool col_ok = true;
        if (MODE == G_FC3)
        {
            osc = tg.q0[n];
            omn = tg.q1[n];
            // columns: channel c occupies [c * MAGP, c * MAGP + 2049); the rest of each half is padding.  A wave's 32
            // consecutive columns are 32 consecutive floats of one output row, line-aligned: whole 128-byte lines per store.
            const int c = n >= MAGP ? 1 : 0, bin = n - c * MAGP;
            col = lo + (unsigned)(c * args.T) * MAGP + (unsigned)bin;
            col_ok = bin < NBINS;
        }
        // running values of the row loop below (incremented row by row: a handful of constants instead of one literal per
        // row, which the compiler would all keep in SGPRs): byte offset of (row, column n) in the block's FIRST lane, and
        // the row's frame index relative to the block's SECOND lane (negative while the row is still in the first)
        unsigned run = (col + (unsigned)fA * MAGP) * 4u;
        int frow = fB;
        // fc3: frames >= T (M padding) and padding columns are dropped by the buffer range check (their byte offset is
        // replaced by one past the end): no branch and no 64-bit address arithmetic per element
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
                            ys[j] = fmaxf(y * osc + omn, 0.f); // inference.cpp:161-166: the mask
                            // (a lane past the launch's last one lands beyond the resource's range and is dropped)
                            offs[j] = ok ? run + (second ? dA : 0u) : 0xfffffff0u;
                            run += MAGP * 4;
                            frow += 1;
                        }
                    }
                }
                if (MODE == G_FC3)
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ys[j]), rs_mag, offs[j], 0, 0);
                    run += 4 * MAGP * 4; // the next group of four rows starts eight rows further
                    frow += 4;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0); // ... and the arithmetic of the next group of rows behind this group's
            }
    }
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


} // namespace umx
