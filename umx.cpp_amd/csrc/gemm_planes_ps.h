// gemm_planes_ps.h -- the ping-pong plane GEMM of gemm_planes_pp.h as a PERSISTENT kernel: one workgroup per CU walks a list of
// 256 x 256 tiles flattened over the four targets, the K loop never drains between tiles, and a tile's epilogue runs INSIDE the first
// trip of the next tile, one 32-row band of the wave tile at a time, each band followed by the matrix instructions that restart its
// accumulators from zero (inference.cpp:86,127,143 and lstm.cpp:132-136: the four dense products of a target).
//
// What a tile of gemm_planes_pp_kernel loses outside its main loop (profiles/r06_pp_tile_profile.txt): the wait for its
// first stage with an empty pipeline, an epilogue of 128 stores per wave whose instruction stream is a chain of table reads, a few
// vector operations and a store per element with nothing to overlap it, the drain of the stores before the workgroup may end, and the
// dispatch of its successor: 14-16 % of a W_ih or fc3 tile, plus ~10 % between workgroups.  Here
//   * the staging stream runs ahead across tile boundaries: when the last trip of a tile is multiplied the first stages of the next
//     tile are already in LDS (ONE tile cursor, decoded STAGES - 1 trips ahead by the staging side and handed on to the multiplying
//     side; the per-piece staging offsets are rebuilt at a tile switch, a trip adds 64 bytes);
//   * stores are fire and forget: every stage in flight is waited for right BEFORE the epilogue's stores go out, so the waits of the
//     two trips behind it are about stages that have landed already and wait for nothing -- the in-order memory counter would
//     otherwise make them count store acknowledgements (the first form relaxed them to "all but the 63 youngest operations", which
//     still waits for 65 acknowledgements: fc3 +2.5 %, -DPS_PREWAIT=0);
//   * the accumulators are the only copy of a tile's result, so its epilogue must be ISSUED before the next tile's first matrix
//     instruction overwrites them -- but not a cycle earlier: trip 0 of a tile handles band mi = 0 .. 3 as { fix-up, bn / bias /
//     activation and stores of the previous tile's band ; the band's matrix instructions of the first 16-k step, starting from a zero
//     C operand }, so the matrix pipe restarts band by band under the vector and store work of the next band.  Only the first 16-k
//     step's fragments are in registers meanwhile (the second step's are read behind the last band): the epilogue's temporaries take
//     the place of those 40 / 48 registers;
//   * per-row factors of the affine fix-up, the rows' byte offsets (fc3: row -> (track lane, frame)) and the per-column vectors of the
//     epilogue are tables in LDS behind the stage buffers: raw by LDS-DMA (no registers) three trips before a tile's end, finished by
//     group 0 in the M phase of the next tile's first trip; the epilogue reads a row group's three vectors one group AHEAD of their use
//     and the column vectors once per tile -- its instruction stream is vector work and stores, not LDS round trips.
// Every accumulator sees the matrix instructions of gemm_planes_pp_kernel in the same order (the order ACROSS accumulators differs
// in trip 0 only) and every output element the same scalar expression: bit-identical results (tests/test_gpu_batch.py).
// Group 1 reads the second half of a first trip's fragments one phase later than the ping-pong schedule has it -- while group 0 is
// in the M phase of trip 1, whose staging would overwrite exactly that stage: group 0 issues trip 1's staging at the START of its C
// phase instead (one phase later; the stage it fills is needed three -- two-stage form: one -- phases on).
#pragma once
#include "gemm_planes_pp.h"

namespace umx
{

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
constexpr int PS_TAB_ROWS = 4, PS_TAB_COLS = 8; // row tables: rsc -> scale, rs0 -> offset, rs1 -> byte offset of the row, rs2; column tables: rm, var -> sd, 1/sd, gw, gb, osc, omn, fc3 column offset
__host__ __device__ constexpr int ps_lds_bytes(int NBP) { return gp_lds_bytes(4, 4, NBP) + (PS_TAB_ROWS + PS_TAB_COLS) * 256 * 4; }
constexpr unsigned PS_DROP = 0x40000000u; // an offset the rebased mask resource (at most two lanes long) never holds
// Cache policy of the epilogue's stores: 2 = nt.  The outputs are streams nobody reads before the kernel has ended (W_ih's P alone is
// 10.8 GB per 64-lane launch): stored non-temporally they do not push the A / B tiles the other workgroups are staging out of the
// L2s.  W_ih 9.65 -> 8.9 ms per launch, fc3 13.7 -> 13.4 (profiles/r06_ps_store_policy.txt; 0 = default policy, 16 = sc1: no change).
#ifndef PS_STORE_AUX
#define PS_STORE_AUX 2
#endif
#ifndef PS_PREWAIT
#define PS_PREWAIT 1 // 1: every stage in flight is waited for BEFORE the epilogue's stores go out, so that no later wait has to count acknowledgements of stores (0: vmcnt(63) behind them)
#endif
#ifndef PS_A_AUX
#define PS_A_AUX 0 // cache policy of the staging loads of A / B (A/B builds)
#endif
#ifndef PS_B_AUX
#define PS_B_AUX 0
#endif
#ifndef PS_PROFILE
#define PS_PROFILE 0 // 1 (timing build): two workgroups print where the cycles of their tiles go
#endif

template <int MODE, int NBP> __global__ __launch_bounds__(512, 1) void gemm_planes_ps_kernel(GemmPArgs args, int ntg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];
    constexpr int MI = 4, BM = 256, BN = 256, KKP = 2;
    constexpr int A_PL = BM * 64, B_PL = BN * 64;
    constexpr int BUF_BYTES = 2 * A_PL + NBP * B_PL, STAGES = gp_stages(4, 4, NBP);
    static_assert(NBP == 1 || NBP == 2, "weight planes: 1 (u8) or 2 (u16, fp32)");
    float *const rowtab = reinterpret_cast<float *>(gp_smem + STAGES * BUF_BYTES); // [4][256]
    float *const coltab = rowtab + PS_TAB_ROWS * 256;                               // [8][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, ws = wave & 3;
    const int wm = grp, wn = ws, lr = lane & 31, lh = lane >> 5;
    const int K = args.K, lda = args.lda, nk = K / GP_BK;

    // ---- this workgroup's tiles: the flat order of gemm_planes_pp_kernel (groups of GEMM_GROUP_M row tiles, column tiles inside)
    // continued over the targets, cut in eight contiguous chunks (one per XCD: workgroup b runs on XCD b % 8), dealt round-robin to
    // the chunk's workgroups
    const int gx = args.N / BN, gy = args.M / BM, total = gx * gy, total_all = total * ntg;
    const int wpx = (int)gridDim.x >> 3, chunk = (total_all + 7) >> 3;
    const int v_begin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3), v_end = min(((int)(blockIdx.x & 7) + 1) * chunk, total_all);
    if (v_begin >= v_end)
        return;
    const int ntiles = (v_end - v_begin + wpx - 1) / wpx, ntrips = ntiles * nk;
    // tile v -> target, first row, first column, and (fc3) the track lane / frame of its first row
#define PS_DECODE(v_, TGI, M0, N0, LN0, F0)                                                                          \
    {                                                                                                                \
        const int tg_ = (v_) / total, vt_ = (v_) - tg_ * total;                                                      \
        const int per_group_ = GEMM_GROUP_M * gx, group_ = vt_ / per_group_, first_m_ = group_ * GEMM_GROUP_M;       \
        const int gsize_ = min(gy - first_m_, GEMM_GROUP_M), in_group_ = vt_ - group_ * per_group_;                  \
        const int q_ = in_group_ / gsize_;                                                                           \
        TGI = tg_;                                                                                                   \
        M0 = (first_m_ + in_group_ - q_ * gsize_) * BM;                                                              \
        N0 = q_ * BN;                                                                                                \
        if (MODE == G_FC3)                                                                                           \
        {                                                                                                            \
            LN0 = M0 / args.Tp_lane;                                                                                 \
            F0 = M0 - LN0 * args.Tp_lane;                                                                            \
        }                                                                                                            \
    }

    // ---- LDS-DMA staging (gemm_planes_pp_kernel's pieces: i = it DEAL + w8, plane i / 16, 16-row group i % 16).  What depends on the
    // wave sits in two scalars (lds_w, the rows 16 w8), what depends on the piece is a compile-time constant or one of the NA + NB
    // kernel constants a_it / b_it, what depends on the tile in a_tile / b_tile, and a trip adds koff.
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const int st_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int voffA = ((lane >> 2) * lda) * 2 + st_chunk * 16, voffB = ((lane >> 2) * K) * 2 + st_chunk * 16;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)gp_smem;
    constexpr bool BOTH = PP_DMA_BOTH && STAGES == 3;
    constexpr int DEAL = BOTH ? 8 : 4, PER = 16 / DEAL; // a wave's pieces per plane
    constexpr int A_GROUPS = 2 * (BM / 16), B_GROUPS = NBP * (BN / 16), NA = A_GROUPS / DEAL, NB = B_GROUPS / DEAL, DMA_PER_WAVE = NA + NB;
    static_assert(DMA_PER_WAVE <= 16 && A_GROUPS % DEAL == 0 && B_GROUPS % DEAL == 0 && BM == 256 && BN == 256, "vmcnt bookkeeping below");
    const int w8 = (BOTH ? 4 * grp : 0) + ws; // 0 .. DEAL - 1
    const bool issuer = grp == 0 || BOTH;
    const unsigned lds_w = lds0 + (unsigned)w8 * 1024u;
    int a_it[NA], b_it[NB];
#pragma unroll
    for (int it = 0; it < NA; ++it)
        a_it[it] = (int)((long)(it / PER) * (long)args.a_plane * 2) + (it % PER) * DEAL * 32 * lda;
#pragma unroll
    for (int it = 0; it < NB; ++it)
        b_it[it] = (int)((long)(it / PER) * (long)args.N * K * 2) + (it % PER) * DEAL * 32 * K;
    // the tile cursor: the trip the next staging batch belongs to; (n_*) = the tile it has entered last, which the multiplying side
    // takes over when it gets there
    int d_tile = 0, d_k = 0, koff = 0, a_tile, b_tile, n_tgi, n_m0, n_n0, n_ln0 = 0, n_f0 = 0;
    PS_DECODE(v_begin, n_tgi, n_m0, n_n0, n_ln0, n_f0)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(args.t[n_tgi].A), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(args.t[n_tgi].B), 0, 0x7fffffff, 0x00020000);
    a_tile = (int)((long)(n_m0 + 16 * w8) * lda * 2);
    b_tile = (int)((long)(n_n0 + 16 * w8) * K * 2);
    int c_tgi = n_tgi, c_m0 = n_m0, c_n0 = n_n0, c_ln0 = n_ln0, c_f0 = n_f0; // the tile being multiplied
// (the scalar offset goes through a local: an array element written straight into the builtin's argument makes the HOST pass drop the
// kernel's definition without a diagnostic -- the library then fails to load with an undefined symbol; ROCm 7.2 clang)
#define PS_DMA_ISSUE(buf)                                                                                            \
    {                                                                                                                \
        const unsigned lb_ = lds_w + (unsigned)((buf)*BUF_BYTES);                                                    \
        const int ao_ = a_tile + koff, bo_ = b_tile + koff;                                                          \
        _Pragma("unroll") for (int it = 0; it < NA; ++it)                                                            \
        {                                                                                                            \
            const int so_ = ao_ + a_it[it];                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(size_t)(lb_ + (it / PER) * A_PL + (it % PER) * DEAL * 1024), 16, voffA, so_, 0, PS_A_AUX); \
        }                                                                                                            \
        _Pragma("unroll") for (int it = 0; it < NB; ++it)                                                            \
        {                                                                                                            \
            const int so_ = bo_ + b_it[it];                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lb_ + 2 * A_PL + (it / PER) * B_PL + (it % PER) * DEAL * 1024), 16, voffB, so_, 0, PS_B_AUX); \
        }                                                                                                            \
    }
#define PS_DMA_ADVANCE()                                                                                             \
    {                                                                                                                \
        koff += 2 * GP_BK;                                                                                           \
        if (++d_k == nk)                                                                                             \
        {                                                                                                            \
            d_k = 0;                                                                                                 \
            koff = 0;                                                                                                \
            if (++d_tile < ntiles)                                                                                   \
            {                                                                                                        \
                PS_DECODE(v_begin + d_tile * wpx, n_tgi, n_m0, n_n0, n_ln0, n_f0)                                    \
                rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(args.t[n_tgi].A), 0, 0x7fffffff, 0x00020000); \
                rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(args.t[n_tgi].B), 0, 0x7fffffff, 0x00020000); \
                a_tile = (int)((long)(n_m0 + 16 * w8) * lda * 2);                                                    \
                b_tile = (int)((long)(n_n0 + 16 * w8) * K * 2);                                                      \
            }                                                                                                        \
        }                                                                                                            \
    }

    floatx16 acc[MI][2];
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int sw = (lr >> 2) & 3;
    const int fragA = (wm * 32 * MI + lr) * 64, fragB = 2 * A_PL + (wn * 64 + lr) * 64;
    f16x8 fa[2][MI][KKP], fb[NBP][2][KKP];
#define PS_LD(off) (*reinterpret_cast<const f16x8 *>(gp_smem + (off)))
#define PS_LOAD_KL(buf, kl)                                                                                          \
    {                                                                                                                \
        const int bo = (buf)*BUF_BYTES;                                                                              \
        const int co = (((kl)*2 + lh) ^ sw) * 16;                                                                    \
        _Pragma("unroll") for (int p = 0; p < NBP; ++p)                                                              \
        {                                                                                                            \
            fb[p][0][kl] = PS_LD(bo + fragB + p * B_PL + co);                                                        \
            fb[p][1][kl] = PS_LD(bo + fragB + p * B_PL + 32 * 64 + co);                                              \
        }                                                                                                            \
        _Pragma("unroll") for (int p = 1; p >= 0; --p)                                                               \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                        \
                fa[p][mi][kl] = PS_LD(bo + fragA + p * A_PL + mi * 32 * 64 + co);                                    \
    }
#define PS_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
    // the terms of one 16-k step for band mi, smallest first (gemm_planes_kernel's order per accumulator); ZERO: the first term starts
    // the accumulators from a zero C operand
#define PS_BAND(mi, kl, ZERO)                                                                                        \
    {                                                                                                                \
        acc[mi][0] = PS_MFMA(fa[1][mi][kl], fb[0][0][kl], (ZERO) ? zero16 : acc[mi][0]);                             \
        acc[mi][1] = PS_MFMA(fa[1][mi][kl], fb[0][1][kl], (ZERO) ? zero16 : acc[mi][1]);                             \
        if (NBP == 2)                                                                                                \
        {                                                                                                            \
            acc[mi][0] = PS_MFMA(fa[0][mi][kl], fb[NBP - 1][0][kl], acc[mi][0]);                                     \
            acc[mi][1] = PS_MFMA(fa[0][mi][kl], fb[NBP - 1][1][kl], acc[mi][1]);                                     \
        }                                                                                                            \
        acc[mi][0] = PS_MFMA(fa[0][mi][kl], fb[0][0][kl], acc[mi][0]);                                               \
        acc[mi][1] = PS_MFMA(fa[0][mi][kl], fb[0][1][kl], acc[mi][1]);                                               \
    }
    // one 16-k step over the four bands in gemm_planes_pp_kernel's instruction order (term by term across the bands)
#define PS_TERM(PA, PB, kl)                                                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                \
    {                                                                                                                \
        acc[mi][0] = PS_MFMA(fa[PA][mi][kl], fb[PB][0][kl], acc[mi][0]);                                             \
        acc[mi][1] = PS_MFMA(fa[PA][mi][kl], fb[PB][1][kl], acc[mi][1]);                                             \
    }
#define PS_STEP(kl)                                                                                                  \
    {                                                                                                                \
        PS_TERM(1, 0, kl)                                                                                            \
        if (NBP == 2)                                                                                                \
            PS_TERM(0, NBP - 1, kl)                                                                                  \
        PS_TERM(0, 0, kl)                                                                                            \
    }
#define PS_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n) & 15) | (((n) >> 4) << 14))
#define PS_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f) // lgkmcnt(0)
#define PS_BARRIER()                                                                                                 \
    {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }

    // ---- the epilogue of the tile whose accumulators are waiting (e_*): fix-up, bn / bias / activation, stores.  Row and column
    // vectors come from the LDS tables; stores go through a resource rebased to the tile.  (The lane's offsets into the tables are
    // recomputed behind an opaque copy of the lane id: left to the compiler they would be hoisted out of the tile loop and held in
    // registers through the main loop, which has none to spare.)
    int e_m0 = 0, e_n0 = 0, e_tgi = 0, e_ln0 = 0, e_f0 = 0;
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(args.t[0].C, 0, 0, 0x00020000);
    const unsigned ldc4 = (unsigned)args.ldc * 4u;
    const float4 *const row4 = reinterpret_cast<const float4 *>(rowtab); // [3][64] groups of four rows
    const uint4 *const row4u = reinterpret_cast<const uint4 *>(rowtab);
    // per tile: the wave's two column vectors (a lane holds column wn 64 + ni 32 + lr of both 32-column halves); per group of four rows
    // (band mi, rq): scale, offset, byte offset -- read one group ahead of their use
    float cv[2][7];
    unsigned cvo[2];
    float4 nx_mu, nx_ad;
    uint4 nx_ro;
    int e_row = 0; // float4 index of the lane's rows of group (band 0, rq 0): wm 32 + lh
#define PS_EPI_ROWS(g_)                                                                                              \
    {                                                                                                                \
        nx_mu = row4[e_row + 2 * (g_)];                                                                              \
        nx_ad = row4[64 + e_row + 2 * (g_)];                                                                         \
        nx_ro = row4u[128 + e_row + 2 * (g_)];                                                                       \
    }
    // rebase the store resource to the waiting tile, read its column vectors and the first row group
#define PS_EPI_SETUP()                                                                                               \
    {                                                                                                                \
        int el_ = lane;                                                                                              \
        asm volatile("" : "+v"(el_));                                                                                \
        e_row = wm * 32 + (el_ >> 5);                                                                                \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                             \
        {                                                                                                            \
            const int cl_ = wn * 64 + ni * 32 + (el_ & 31);                                                          \
            cvo[ni] = (unsigned)cl_ * 4u;                                                                            \
            cv[ni][4] = coltab[4 * 256 + cl_];                                                                       \
            if (MODE != G_IH)                                                                                        \
            {                                                                                                        \
                cv[ni][0] = coltab[cl_];                                                                             \
                cv[ni][1] = coltab[256 + cl_];                                                                       \
                cv[ni][2] = coltab[2 * 256 + cl_];                                                                   \
                cv[ni][3] = coltab[3 * 256 + cl_];                                                                   \
            }                                                                                                        \
            if (MODE == G_FC3)                                                                                       \
            {                                                                                                        \
                cv[ni][5] = coltab[5 * 256 + cl_];                                                                   \
                cv[ni][6] = coltab[6 * 256 + cl_];                                                                   \
                cvo[ni] = reinterpret_cast<const unsigned *>(coltab)[7 * 256 + cl_];                                 \
            }                                                                                                        \
        }                                                                                                            \
        PS_EPI_ROWS(0)                                                                                               \
        if (MODE == G_FC3)                                                                                           \
        {                                                                                                            \
            const size_t lane_b_ = args.mag_lane * 4, all_b_ = (size_t)(args.lanes ? args.lanes : args.M / args.Tp_lane) * lane_b_; \
            const size_t lo_ = (size_t)e_ln0 * lane_b_, rest_ = all_b_ > lo_ ? all_b_ - lo_ : 0;                      \
            rsC = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(args.t[e_tgi].C) + lo_, 0,     \
                                                    (int)(rest_ < 2 * lane_b_ ? rest_ : 2 * lane_b_), 0x00020000);   \
        }                                                                                                            \
        else                                                                                                         \
            rsC = __builtin_amdgcn_make_buffer_rsrc(args.t[e_tgi].C + ((size_t)e_m0 * args.ldc + e_n0), 0, 0x7fffffff, 0x00020000); \
    }
    // one band = four groups of four rows x 64 columns.  A group's EIGHT elements (4 rows x the lane's 2 columns) are computed side by
    // side and stored together: with one element at a time a wave alone on its SIMD (the partner group waits at the phase barrier)
    // runs a chain of dependent vector operations and a store per element at the latency of each (profiles/r06_ps_tile_profile.txt)
#define PS_EPI_BAND(mi)                                                                                              \
    _Pragma("unroll") for (int rq = 0; rq < 4; ++rq)                                                                 \
    {                                                                                                                \
        const float mus_[4] = {nx_mu.x, nx_mu.y, nx_mu.z, nx_mu.w}, ads_[4] = {nx_ad.x, nx_ad.y, nx_ad.z, nx_ad.w};  \
        const unsigned ro_[4] = {nx_ro.x, nx_ro.y, nx_ro.z, nx_ro.w};                                                \
        if ((mi)*4 + rq + 1 < 16)                                                                                    \
            PS_EPI_ROWS((mi)*4 + rq + 1)                                                                             \
        float ys_[2][4];                                                                                             \
        unsigned os_[2][4];                                                                                          \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
            {                                                                                                        \
                float y = mus_[j] * acc[mi][ni][4 * rq + j] + ads_[j];                                               \
                if (MODE == G_IH)                                                                                    \
                    y = y + cv[ni][4]; /* lstm.cpp:132-135: W_ih x + b_ih */                                         \
                else                                                                                                 \
                {                                                                                                    \
                    y = div_by(y - cv[ni][0], cv[ni][1], cv[ni][2]) * cv[ni][3] + cv[ni][4]; /* batchnorm, inference.cpp:93-97 order */ \
                    if (MODE == G_FC1)                                                                               \
                        y = tanh_epi(y);                                                                             \
                    else if (MODE == G_FC2)                                                                          \
                        y = fmaxf(y, 0.f);                                                                           \
                    else                                                                                             \
                        y = fmaxf(y * cv[ni][5] + cv[ni][6], 0.f); /* inference.cpp:161-166: the mask */             \
                }                                                                                                    \
                ys_[ni][j] = y;                                                                                      \
                os_[ni][j] = ro_[j] + cvo[ni];                                                                       \
            }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0); /* the eight elements' arithmetic in front of their stores, whatever order it takes */ \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ys_[ni][j]), rsC, os_[ni][j], 0, PS_STORE_AUX);   \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }

    // ---- the tables of a tile.  RAW: trip nk - 3 of tile (c_*) drops the tile's rows of rsc / rs0 / rs1 / rs2 and its columns of the
    // epilogue vectors into LDS by the same LDS-DMA as the stages (4 bytes per lane: no registers; issued in front of the trip's staging
    // batch, so that the waits that follow cover them).  FINISH: in the M phase of the next tile's first trip -- where half of the
    // fragment registers are free -- the threads of group 0 turn one row and one column each into what the epilogue reads: the two
    // factors of the affine fix-up, the row's byte offset, sqrt(var + eps) and its reciprocal, fc3's column offset.
    const unsigned lds_row = lds0 + STAGES * BUF_BYTES, lds_col = lds_row + PS_TAB_ROWS * 1024;
#define PS_TAB_PIECE(PTR, LDSOFF, FIRST_ELEM)                                                                        \
    {                                                                                                                \
        if (PTR)                                                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(PTR), 0, 0x7fffffff, 0x00020000), \
                                                     (lds_ptr)(size_t)((LDSOFF) + ws * 256), 4, lane * 4, ((FIRST_ELEM) + ws * 64) * 4, 0, 0); \
    }
#define PS_TAB_RAW()                                                                                                 \
    {                                                                                                                \
        const GemmPTarget &tg_ = args.t[c_tgi];                                                                      \
        if (!BOTH || grp == 0)                                                                                       \
        {                                                                                                            \
            PS_TAB_PIECE(tg_.rsc, lds_row, c_m0)                                                                     \
            PS_TAB_PIECE(tg_.rs0, lds_row + 1024, c_m0)                                                              \
            PS_TAB_PIECE(tg_.rs1, lds_row + 2048, c_m0)                                                              \
            PS_TAB_PIECE(tg_.rs2, lds_row + 3072, c_m0)                                                              \
        }                                                                                                            \
        if (!BOTH || grp == 1)                                                                                       \
        {                                                                                                            \
            if (MODE == G_IH)                                                                                        \
                PS_TAB_PIECE(tg_.e0, lds_col + 4 * 1024, c_n0)                                                       \
            else                                                                                                     \
            {                                                                                                        \
                PS_TAB_PIECE(tg_.e0, lds_col, c_n0)                                                                  \
                PS_TAB_PIECE(tg_.e1, lds_col + 1024, c_n0)                                                           \
                PS_TAB_PIECE(tg_.e2, lds_col + 3 * 1024, c_n0)                                                       \
                PS_TAB_PIECE(tg_.e3, lds_col + 4 * 1024, c_n0)                                                       \
            }                                                                                                        \
            if (MODE == G_FC3)                                                                                       \
            {                                                                                                        \
                PS_TAB_PIECE(tg_.q0, lds_col + 5 * 1024, c_n0)                                                       \
                PS_TAB_PIECE(tg_.q1, lds_col + 6 * 1024, c_n0)                                                       \
            }                                                                                                        \
        }                                                                                                            \
    }
#define PS_TAB_FINISH()                                                                                              \
    {                                                                                                                \
        if (grp == 0)                                                                                                \
        {                                                                                                            \
            const GemmPTarget &tg_ = args.t[e_tgi];                                                                  \
            const int sel_ = e_n0 >= tg_.bsplit ? 1 : 0;                                                             \
            const float rsc_ = tg_.rsc ? rowtab[tid] : args.a_unscale, r0_ = rowtab[256 + tid];                      \
            const float r1_ = tg_.rs1 ? rowtab[512 + tid] : 0.f, r2_ = tg_.rs2 ? rowtab[768 + tid] : 0.f;            \
            unsigned off_ = (unsigned)tid * ldc4; /* byte offset of the row in the rebased C */                       \
            if (MODE == G_FC3)                                                                                       \
            {                                                                                                        \
                /* row -> (track lane, frame): lanes follow each other every Tp_lane rows (>= the tile height), so a tile holds rows \
                   of at most two lanes; frames >= T (M padding) are dropped */                                      \
                const int f1_ = e_f0 + tid, second_ = f1_ >= args.Tp_lane ? 1 : 0, f_ = f1_ - second_ * args.Tp_lane; \
                off_ = f_ < args.T ? ((unsigned)second_ * (unsigned)args.mag_lane + (unsigned)f_ * MAGP) * 4u : PS_DROP; \
            }                                                                                                        \
            rowtab[tid] = tg_.bs[sel_] * rsc_; /* a power of two times s: exact scaling */                           \
            rowtab[256 + tid] = tg_.bo2[sel_] * (tg_.rs2 ? (r0_ + r1_) + r2_ : tg_.rs1 ? r0_ + r1_ : r0_);           \
            reinterpret_cast<unsigned *>(rowtab)[512 + tid] = off_;                                                  \
            if (MODE != G_IH)                                                                                        \
            {                                                                                                        \
                const float sd_ = sqrtf(coltab[256 + tid] + 1e-5f); /* inference.cpp:94-95 */                        \
                coltab[256 + tid] = sd_;                                                                             \
                coltab[2 * 256 + tid] = 1.0f / sd_;                                                                  \
            }                                                                                                        \
            if (MODE == G_FC3)                                                                                       \
            {                                                                                                        \
                /* columns: channel c occupies [c MAGP, c MAGP + 2049); the rest of each half is padding */          \
                const int n_ = e_n0 + tid, ch_ = n_ >= MAGP ? 1 : 0, bin_ = n_ - ch_ * MAGP;                         \
                reinterpret_cast<unsigned *>(coltab)[7 * 256 + tid] = bin_ < NBINS ? ((unsigned)(ch_ * args.T) * MAGP + (unsigned)bin_) * 4u : PS_DROP; \
            }                                                                                                        \
        }                                                                                                            \
    }

    // ---- the phases of a trip.  M: staging of trip u + STAGES - 1, the fragments of this trip (FIRST trip of a tile: of its first
    // 16-k step only).  LATE: group 0 stages at the start of its C phase (kt = 1: group 1 is still reading trip u - 1's stage).
    // YOUNG: the wait of a trip behind an epilogue -- about a stage that landed before the epilogue's stores were issued (PS_PREWAIT).
    int cur = 0, u = 0;
#define PS_M(FIRST, LATE, YOUNG1, TAB, FINISH, SPLIT)                                                                       \
    {                                                                                                                \
        const int nxt_ = STAGES == 3 ? (cur == 0 ? 2 : cur - 1) : cur ^ 1; /* stage of trip u + STAGES - 1 = stage of trip u - 1 */ \
        const bool more_ = u + STAGES - 1 < ntrips, late_ = (LATE) && grp == 0;                                      \
        if ((TAB) && issuer)                                                                                         \
            PS_TAB_RAW()                                                                                             \
        if (more_ && !late_)                                                                                         \
        {                                                                                                            \
            if (issuer)                                                                                              \
                PS_DMA_ISSUE(nxt_)                                                                                   \
            PS_DMA_ADVANCE()                                                                                         \
        }                                                                                                            \
        PS_LOAD_KL(cur, 0)                                                                                           \
        if (!(FIRST))                                                                                                \
            PS_LOAD_KL(cur, 1)                                                                                       \
        if (FINISH)                                                                                                  \
            PS_TAB_FINISH()                                                                                          \
        PS_WAIT_LDS();                                                                                               \
        if (BOTH && grp == 1 && u + 1 < ntrips)                                                                      \
        {                                                                                                            \
            /* group 1's half of trip u + 1 has landed; the pieces of trip u + 2 it has just issued may stay in flight */ \
            if (YOUNG1)                                                                                              \
            {                                                                                                        \
                if (!PS_PREWAIT)                                                                                     \
                    PS_WAIT_VM(63);                                                                                  \
            }                                                                                                        \
            else if (u + 2 < ntrips)                                                                                 \
                PS_WAIT_VM(DMA_PER_WAVE);                                                                            \
            else                                                                                                     \
                PS_WAIT_VM(0);                                                                                       \
        }                                                                                                            \
        if (!(SPLIT) || grp == 0) /* SPLIT: group 1's barrier follows its epilogue, see the tile loop */             \
            PS_BARRIER()                                                                                             \
        if (more_ && late_)                                                                                          \
        {                                                                                                            \
            PS_DMA_ISSUE(nxt_)                                                                                       \
            PS_DMA_ADVANCE()                                                                                         \
        }                                                                                                            \
    }
    // end of a C phase: trip u + 1 (read from the next phase on) has landed; trip u + 2's batch may stay in flight
#define PS_C_END(YOUNG0)                                                                                             \
    {                                                                                                                \
        if (grp == 0 && u + 1 < ntrips)                                                                              \
        {                                                                                                            \
            if (YOUNG0)                                                                                              \
            {                                                                                                        \
                if (!PS_PREWAIT)                                                                                     \
                    PS_WAIT_VM(63);                                                                                  \
            }                                                                                                        \
            else if (STAGES == 3 && u + 2 < ntrips)                                                                  \
                PS_WAIT_VM(DMA_PER_WAVE);                                                                            \
            else                                                                                                     \
                PS_WAIT_VM(0);                                                                                       \
        }                                                                                                            \
        PS_BARRIER()                                                                                                 \
        cur = cur + 1 == STAGES ? 0 : cur + 1;                                                                       \
        ++u;                                                                                                         \
    }
    // the second 16-k step of a tile's first trip: its fragments are read here, behind the bands
#define PS_C_FIRST_TAIL()                                                                                            \
    {                                                                                                                \
        PS_LOAD_KL(cur, 1)                                                                                           \
        PS_WAIT_LDS();                                                                                               \
        PS_STEP(1)                                                                                                   \
    }

    // ---- prologue: the first STAGES - 1 stages
    const int tab_trip = nk - 3;
    {
        if (issuer)
            PS_DMA_ISSUE(0)
        PS_DMA_ADVANCE()
        if (STAGES == 3)
        {
            if (issuer)
                PS_DMA_ISSUE(1)
            PS_DMA_ADVANCE()
            if (issuer)
                PS_WAIT_VM(DMA_PER_WAVE);
        }
        else if (issuer)
            PS_WAIT_VM(0);
    }
    PS_BARRIER() // trip 0 is there
    if (grp == 1)
        PS_BARRIER() // group 1 runs one phase behind
    // ---- first trip of the first tile: nothing to store yet
    PS_M(true, false, false, false, false, false)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
        PS_BAND(mi, 0, true)
    PS_C_FIRST_TAIL()
    __builtin_amdgcn_s_setprio(0);
    PS_C_END(false)
    [[maybe_unused]] long long pq_loop = 0, pq_epi = 0, pq_epi_m = 0, pq_t0 = 0, pq_t1 = 0, pq_t2 = 0;
    const long long pq_start = PS_PROFILE ? clock64() : 0;
    for (int tile = 0; tile < ntiles; ++tile)
    {
        if (PS_PROFILE)
            pq_t0 = clock64();
        for (int kt = 1; kt < nk; ++kt)
        {
            const bool tab = kt == tab_trip;
            PS_M(false, kt == 1, tile > 0 && kt == 1, tab, false, false)
            __builtin_amdgcn_s_setprio(1);
            PS_STEP(0)
            PS_STEP(1)
            __builtin_amdgcn_s_setprio(0);
            PS_C_END(tile > 0 && kt == 1 && STAGES == 3)
        }
        if (PS_PROFILE)
        {
            pq_t1 = clock64();
            pq_loop += pq_t1 - pq_t0;
        }
        e_m0 = c_m0;
        e_n0 = c_n0;
        e_tgi = c_tgi;
        e_ln0 = c_ln0;
        e_f0 = c_f0;
        if (tile + 1 < ntiles)
        {
            // ---- first trip of the next tile (the staging cursor entered it STAGES - 1 trips ago), the waiting tile's epilogue inside:
            // band by band { fix-up, activation, stores ; the band's first 16-k step from a zero C operand } -- the matrix pipe restarts
            // under the next band's vector and store work
            c_m0 = n_m0;
            c_n0 = n_n0;
            c_tgi = n_tgi;
            c_ln0 = n_ln0;
            c_f0 = n_f0;
            // Both groups run the epilogue in the SAME phase -- group 0 in its C phase, group 1 at the end of its M phase (its barrier
            // follows the epilogue instead of preceding it): the CU's store path (32 bytes a clock: a tile's 256 KB take ~8,000 cycles)
            // and the vector pipes serve eight waves at once, and the trip has one long phase instead of two.
            PS_M(true, false, false, false, true, true)
            if (PS_PROFILE)
            {
                pq_t2 = clock64();
                pq_epi_m += pq_t2 - pq_t1;
            }
            if (PS_PREWAIT && issuer)
                PS_WAIT_VM(0); // the stages of trips u + 1 (and u + 2) have landed: the waits of the next two trips are about nothing younger
            PS_EPI_SETUP()
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                PS_EPI_BAND(mi)
            if (grp == 1)
                PS_BARRIER()
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                PS_BAND(mi, 0, true)
            PS_C_FIRST_TAIL()
            __builtin_amdgcn_s_setprio(0);
            PS_C_END(true)
            if (PS_PROFILE)
                pq_epi += clock64() - pq_t2;
        }
    }
    if (PS_PROFILE && (blockIdx.x == 8 || blockIdx.x == 101) && (tid == 0 || tid == 256))
        printf("# ps<%d,%d> wg %d wave %d tiles %d trips/tile %d: cycles  normal trip %lld  first trip: M phase %lld, C phase with the epilogue %lld  per tile %lld\n", MODE, NBP,
               (int)blockIdx.x, wave, ntiles, nk, pq_loop / ((long long)ntiles * (nk - 1)), pq_epi_m / max(ntiles - 1, 1), pq_epi / max(ntiles - 1, 1),
               (clock64() - pq_start) / ntiles);
    if (grp == 0)
        PS_BARRIER() // group 1's last C phase
    // ---- the last tile's epilogue
    PS_TAB_FINISH()
    PS_BARRIER()
    PS_EPI_SETUP()
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
        PS_EPI_BAND(mi)
#undef PS_M
#undef PS_C_END
#undef PS_C_FIRST_TAIL
#undef PS_DECODE
#undef PS_DMA_ISSUE
#undef PS_DMA_ADVANCE
#undef PS_LD
#undef PS_LOAD_KL
#undef PS_MFMA
#undef PS_BAND
#undef PS_TERM
#undef PS_STEP
#undef PS_WAIT_VM
#undef PS_WAIT_LDS
#undef PS_BARRIER
#undef PS_EPI_ROWS
#undef PS_EPI_BAND
#undef PS_EPI_SETUP
#undef PS_TAB_PIECE
#undef PS_TAB_RAW
#undef PS_TAB_FINISH
}

} // namespace umx
