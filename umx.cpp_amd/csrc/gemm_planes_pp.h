// gemm_planes_pp.h -- the 256 x 256 plane GEMM of gemm_planes.h with its eight waves in PING-PONG: the two waves of a
// SIMD take turns at the matrix pipe.
//
// What gemm_planes_kernel's main loop loses (profiles/r03_gemm_pace_experiments.txt, GP_PROFILE): all waves leave the
// K tile's barrier together, read their fragments together (the matrix pipe idle meanwhile), then queue their matrix
// instructions together -- a trip takes ~2,950 cycles where its matrix instructions need 2,048.  (The LDS port itself is not
// the limit: 256 B/clk/CU for exactly these reads, tools/lds_probe.hip, of which a trip's reads + DMA writes take 40 %; what
// is, is the L2 -> LDS path and the epilogue: DESIGN 4.6, "Round 4: one diagnosis".)  Here the
// waves are two groups of four (one wave of each group per SIMD: waves are dealt to the SIMDs round-robin), half a
// trip apart:
//
//      phase      2i            2i+1          2i+2          2i+3
//      group 0    M(i)     |    C(i)     |    M(i+1)   |    C(i+1)   |        M = DMA issue + ALL fragment reads of
//      group 1    C(i-1)   |    M(i)     |    C(i)     |    M(i+1)   |            one unit into registers
//                          ^ s_barrier (all eight waves) at every phase boundary      C = that unit's 32 matrix instructions
//
// so that at any time one wave of a SIMD owns the matrix pipe while the other reads the fragments of its next unit
// (20 KB per wave: 640 cycles of the LDS port for the four reading waves against 1,024 cycles of matrix instructions).
// A unit is a whole 32-k tile: 32 matrix instructions and 80 fragment registers for one-plane (u8) weights, 48 and 96 for
// two-plane (u16) ones, beside the 128 accumulator registers of the 128 x 64 wave tile.  Group 0 issues all LDS-DMA (in its M phase, into the stage both groups finished reading one barrier ago) and
// waits for a tile at the end of the C phase before the M phase that reads it: a tile has two trips to arrive, as in
// gemm_planes_kernel.  Every accumulator sees the same sequence of matrix instructions as there: bit-identical results.
#pragma once
#include "gemm_planes.h"

#ifndef PP_DMA_BOTH
#define PP_DMA_BOTH 1 // 0: group 0 issues every piece of the staging (the form of round 3; A/B builds)
#endif

namespace umx
{

template <int MODE, int NBP> __global__ __launch_bounds__(512, 1) void gemm_planes_pp_kernel(GemmPArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];
    constexpr int MI = 4, BM = 256, BN = 256;
    constexpr int A_PL = BM * 64, B_PL = BN * 64; // bytes of one plane tile of each operand
    constexpr int BUF_BYTES = 2 * A_PL + NBP * B_PL, STAGES = gp_stages(4, 4, NBP);
    constexpr int PH = 1, KKP = 2 / PH; // phases per K tile, 16-k steps per phase (half-tile phases for the two-plane weights
                                        // -- 32 matrix instructions per phase like the one-plane form -- measured slower: four
                                        // barriers per trip; fc2 4.9 against 4.6 ms, fc3 9.4 against 9.0)
    static_assert(NBP == 1 || NBP == 2, "weight planes: 1 (u8) or 2 (u16, fp32)");
    const GemmPTarget tg = args.t[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, ws = wave & 3; // group (phase offset), SIMD
    const int wm = grp, wn = ws, lr = lane & 31, lh = lane >> 5;
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

    // ---- LDS-DMA staging: as gemm_planes_kernel, the 16-row groups dealt to the four waves of group 0
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(tg.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(tg.B), 0, 0x7fffffff, 0x00020000);
    const int st_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int voffA = ((lane >> 2) * lda) * 2 + st_chunk * 16, voffB = ((lane >> 2) * K) * 2 + st_chunk * 16;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)gp_smem;
    const long a_plane_b = (long)args.a_plane * 2, b_plane_b = (long)args.N * K * 2;
    // BOTH (three stages = one-plane weights): each group issues HALF of a tile's pieces, in its own M phase.  With all of
    // them in group 0's M phase the L2 -> LDS path (31 B/clk/CU with every CU streaming, tools/lds_probe mode 3: 48 KB need
    // ~1,570 cycles) has one 1,024-cycle phase to take a whole tile and none in the next: the phase stretches to the path's
    // pace and the partner's matrix instructions wait at the barrier (profiles/r04_pp_pace_experiments.txt: W_ih 4.9 ms with
    // the staging, 3.6 without).  Dealt to both phases the path is busy all the time at 24 KB per phase.
    constexpr bool BOTH = PP_DMA_BOTH && STAGES == 3;
    constexpr int DEAL = BOTH ? 8 : 4; // pieces i = it DEAL + deal0 + SIMD: dealt to the eight waves, or to group 0's four
    constexpr int A_GROUPS = 2 * (BM / 16), B_GROUPS = NBP * (BN / 16), DMA_PER_WAVE = (A_GROUPS + B_GROUPS) / DEAL;
    static_assert(DMA_PER_WAVE <= 16 && A_GROUPS % DEAL == 0 && B_GROUPS % DEAL == 0, "vmcnt bookkeeping below");
    const int deal0 = BOTH ? 4 * grp : 0;
#define PP_DMA(buf, k0)                                                                                              \
    {                                                                                                                \
        _Pragma("unroll") for (int it = 0; it < A_GROUPS / DEAL; ++it)                                               \
        {                                                                                                            \
            const int i = it * DEAL + deal0 + ws, p = i / (BM / 16), j = i % (BM / 16);                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(size_t)(lds0 + (buf)*BUF_BYTES + p * A_PL + j * 1024), 16, voffA, \
                                                     (int)(p * a_plane_b + ((long)(m0 + 16 * j) * lda + (k0)) * 2), 0, 0); \
        }                                                                                                            \
        _Pragma("unroll") for (int it = 0; it < B_GROUPS / DEAL; ++it)                                               \
        {                                                                                                            \
            const int i = it * DEAL + deal0 + ws, p = i / (BN / 16), j = i % (BN / 16);                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lds0 + (buf)*BUF_BYTES + 2 * A_PL + p * B_PL + j * 1024), 16, voffB, \
                                                     (int)(p * b_plane_b + ((long)(n0 + 16 * j) * K + (k0)) * 2), 0, 0); \
        }                                                                                                            \
    }

    floatx16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            acc[mi][0][r] = 0.f;
            acc[mi][1][r] = 0.f;
        }
    const int sw = (lr >> 2) & 3;
    const int fragA = (wm * 32 * MI + lr) * 64, fragB = 2 * A_PL + (wn * 64 + lr) * 64;
    f16x8 fa[2][MI][KKP], fb[NBP][2][KKP]; // the fragments of one unit: 16 + 4 (u8) or 8 + 4 (u16) x 4 registers
#define PP_LD(off) (*reinterpret_cast<const f16x8 *>(gp_smem + (off)))
#define PP_LOAD(buf, ph)                                                                                             \
    {                                                                                                                \
        const int bo = (buf)*BUF_BYTES;                                                                              \
        _Pragma("unroll") for (int kl = 0; kl < KKP; ++kl)                                                           \
        {                                                                                                            \
            const int co = ((((ph)*KKP + kl) * 2 + lh) ^ sw) * 16;                                                   \
            _Pragma("unroll") for (int p = 0; p < NBP; ++p)                                                          \
            {                                                                                                        \
                fb[p][0][kl] = PP_LD(bo + fragB + p * B_PL + co);                                                    \
                fb[p][1][kl] = PP_LD(bo + fragB + p * B_PL + 32 * 64 + co);                                          \
            }                                                                                                        \
            _Pragma("unroll") for (int p = 1; p >= 0; --p)                                                           \
                _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                    \
                    fa[p][mi][kl] = PP_LD(bo + fragA + p * A_PL + mi * 32 * 64 + co);                                \
        }                                                                                                            \
    }
#define PP_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
#define PP_TERM(PA, PB, kl)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                \
    {                                                                                                                \
        PP_MFMA(fa[PA][mi][kl], fb[PB][0][kl], acc[mi][0]) PP_MFMA(fa[PA][mi][kl], fb[PB][1][kl], acc[mi][1])        \
    }
    // smallest terms first: the order of gemm_planes_kernel
#define PP_MMA()                                                                                                     \
    _Pragma("unroll") for (int kl = 0; kl < KKP; ++kl)                                                               \
    {                                                                                                                \
        if (NBP == 1)                                                                                                \
        {                                                                                                            \
            PP_TERM(1, 0, kl) PP_TERM(0, 0, kl)                                                                      \
        }                                                                                                            \
        else                                                                                                         \
        {                                                                                                            \
            PP_TERM(1, 0, kl) PP_TERM(0, NBP - 1, kl) PP_TERM(0, 0, kl)                                              \
        }                                                                                                            \
    }
    // vmcnt(n): all but the n most recent DMA instructions of this wave have landed (n <= 63: bits 3:0 and 15:14)
#define PP_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14))
#define PP_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f) // lgkmcnt(0)
    // phase boundary: nothing is scheduled across it
#define PP_BARRIER()                                                                                                 \
    {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }

// timing builds only (garbage results), tools/build_variants.sh: 1 = no staging inside the main loop, 2 = no matrix
// instructions, 4 = no fragment reads, 8 = no epilogue (stores of row 0 only) -- which side sets the pace of the shipped loop
#ifndef PP_EXPERIMENT
#define PP_EXPERIMENT 0
#endif
#ifndef PP_PROFILE
#define PP_PROFILE 0 // 1 (timing build): one workgroup of the launch's second round prints where the cycles of its tile go
#endif
    [[maybe_unused]] long long pq[6] = {0, 0, 0, 0, 0, 0};
    if (PP_PROFILE)
        pq[0] = clock64();
    const int nk = K / GP_BK;
    if (grp == 0 || BOTH)
    {
        PP_DMA(0, 0)
        if (STAGES == 3 && nk > 1)
        {
            PP_DMA(1, GP_BK)
            PP_WAIT_VM(DMA_PER_WAVE);
        }
        else
            PP_WAIT_VM(0);
    }
    PP_BARRIER() // tile 0 is there
    if (PP_PROFILE)
        pq[1] = clock64();
    if (grp == 1)
        PP_BARRIER() // group 1 runs one phase behind
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt)
    {
        const int nxt = STAGES == 3 ? (cur == 0 ? 2 : cur - 1) : cur ^ 1; // stage of tile kt + STAGES - 1 = stage of tile kt - 1
#pragma unroll
        for (int ph = 0; ph < PH; ++ph)
        {
            // M: the stage of tile kt - 1 was last read by group 1 one phase ago
            if (!(PP_EXPERIMENT & 1) && ph == 0 && (grp == 0 || BOTH) && kt + STAGES - 1 < nk)
                PP_DMA(nxt, (kt + STAGES - 1) * GP_BK)
            if (!(PP_EXPERIMENT & 4) || kt == 0)
                PP_LOAD(cur, ph)
            PP_WAIT_LDS();
            if (BOTH && grp == 1 && kt + 1 < nk)
            {
                // group 1's half of tile kt + 1 (issued one trip ago; group 0 reads the tile from the next phase on) has landed;
                // the pieces of tile kt + 2 it has just issued may stay in flight
                if (kt + 2 < nk)
                    PP_WAIT_VM(DMA_PER_WAVE);
                else
                    PP_WAIT_VM(0);
            }
            PP_BARRIER()
            // C
            __builtin_amdgcn_s_setprio(1);
            if (!(PP_EXPERIMENT & 2) || kt == 0)
                PP_MMA()
            __builtin_amdgcn_s_setprio(0);
            if (ph == PH - 1 && grp == 0 && kt + 1 < nk)
            {
                // tile kt + 1 (read from the next phase on) has landed; tile kt + 2's batch may stay in flight
                if (STAGES == 3 && kt + 2 < nk)
                    PP_WAIT_VM(DMA_PER_WAVE);
                else
                    PP_WAIT_VM(0);
            }
            PP_BARRIER()
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
    if (grp == 0)
        PP_BARRIER() // group 1's last C phase
    if (PP_PROFILE)
        pq[2] = clock64();
#undef PP_DMA
#undef PP_LD
#undef PP_LOAD
#undef PP_MFMA
#undef PP_TERM
#undef PP_MMA
#undef PP_WAIT_VM
#undef PP_WAIT_LDS
#undef PP_BARRIER
    {
        // the affine fix-up and the epilogue of gemm_planes_kernel (the stage buffers are free now)
        const int sel = n0 >= tg.bsplit ? 1 : 0;
        const float bsc = tg.bs[sel], o2 = tg.bo2[sel];
        float *const fx = reinterpret_cast<float *>(gp_smem); // [2][BM]
        __syncthreads();
        for (int i = tid; i < BM; i += 512)
        {
            const int m = m0 + i;
            fx[i] = bsc * (tg.rsc ? tg.rsc[m] : args.a_unscale);
            fx[BM + i] = o2 * (tg.rs2 ? (tg.rs0[m] + tg.rs1[m]) + tg.rs2[m] : tg.rs1 ? tg.rs0[m] + tg.rs1[m] : tg.rs0[m]);
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
            {
                const int ml = wm * 32 * MI + mi * 32 + 8 * rq + 4 * lh;
                const float4 mu = *reinterpret_cast<const float4 *>(fx + ml), ad = *reinterpret_cast<const float4 *>(fx + BM + ml);
                const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, ads[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int r = 4 * rq + j;
                    acc[mi][0][r] = mus[j] * acc[mi][0][r] + ads[j];
                    acc[mi][1][r] = mus[j] * acc[mi][1][r] + ads[j];
                }
                asm volatile("" ::: "memory");
            }
    }
    if (PP_PROFILE)
        pq[3] = clock64();
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
    if (PP_EXPERIMENT & 8)
    {
        float s = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s += acc[mi][0][r] + acc[mi][1][r];
        if (s == 12345.678f)
            tg.C[tid] = s;
        return;
    }
#pragma unroll
    for (int half = 0; half < MI / 2; ++half)
        gemm_epilogue<MODE>(et, ea, m0 + wm * 32 * MI + half * 64, n0, 0, wn, lr, lh, acc[2 * half][0], acc[2 * half][1], acc[2 * half + 1][0],
                            acc[2 * half + 1][1]);
    if (PP_PROFILE)
    {
        pq[4] = clock64();
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): the stores are acknowledged
        pq[5] = clock64();
        if ((blockIdx.x == 264 || blockIdx.x == 2000) && blockIdx.z == 1 && (tid == 0 || tid == 64 * 4))
            printf("# pp<%d,%d> wg %d wave %d trips %d: cycles  first-tile-wait %lld  main-loop %lld (%lld per trip)  fix-up %lld  epilogue-issue %lld  store-drain %lld  total %lld\n",
                   MODE, NBP, (int)blockIdx.x, wave, nk, pq[1] - pq[0], pq[2] - pq[1], (pq[2] - pq[1]) / nk, pq[3] - pq[2], pq[4] - pq[3], pq[5] - pq[4], pq[5] - pq[0]);
    }
}

} // namespace umx
