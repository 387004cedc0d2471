// tools/i8_loop_probe.hip -- TIMING probe (results are not checked) of the main loop planned for the int8 form of the dense
// stack: 256 x 128 block, eight waves of 64 x 64 in ping-pong (csrc/gemm_planes_pp.h's schedule), three int8 digit
// planes of A against one int8 plane of B, v_mfma_i32_32x32x32_i8 into one accumulator set per digit, K-tiled operands
// ([K/32][rows][32 B]: a wave-instruction of the LDS-DMA fetches 32 consecutive rows = 1 KiB contiguous), FIVE 28 KB stages
// (four tiles in flight).  No epilogue: the accumulators are folded into one word per lane so that nothing is optimised away.
// Prints the time per tile and CU, what that is per 32-k trip (768 matrix cycles per SIMD) and the L2 -> LDS rate.
// -DPROBE_BM=128: the three-digit operand on the SHORT side of the block (128 x 256: 20 KB per stage instead of 28).
//   hipcc --offload-arch=gfx950 -O3 -o tools/i8_loop_probe tools/i8_loop_probe.hip && tools/i8_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

#ifndef PROBE_BM
#define PROBE_BM 256 // rows of the three-digit operand per block: 256 (x 128 columns) or 128 (x 256 columns)
#endif
constexpr int BM = PROBE_BM, BN = 256 * 128 / PROBE_BM, BKB = 32; // block tile; bytes (= int8 k) per row and stage
#ifndef PROBE_STAGES
#define PROBE_STAGES 5
#endif
constexpr int A_PL = BM * BKB, B_PL = BN * BKB, STAGE = 3 * A_PL + B_PL, STAGES = PROBE_STAGES;
static_assert(STAGES * STAGE <= 160 * 1024 && STAGES >= 2 && STAGES <= 8, "LDS");
constexpr int A_GROUPS = 3 * (BM / 32), B_GROUPS = BN / 32;
constexpr int DMA_PER_WAVE = (A_GROUPS + B_GROUPS) / 4; // 28 (256 x 128) or 20 (128 x 256) wave-instructions per tile, dealt to the four waves of group 0
static_assert(A_GROUPS % 4 == 0 && B_GROUPS % 4 == 0, "dealt evenly");

#define CHECK(e)                                                                                      \
    do                                                                                                \
    {                                                                                                 \
        hipError_t _e = (e);                                                                          \
        if (_e != hipSuccess)                                                                         \
        {                                                                                             \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                                   \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

// A: [3 digits][K/32][M][32] int8, B: [K/32][N][32] int8, out: one int per thread
__global__ __launch_bounds__(512, 1) void i8_loop_kernel(const signed char *A, const signed char *B, int *out, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, ws = wave & 3, lr = lane & 31, lh = lane >> 5;
    const int gx = N / BN, tile = blockIdx.x, tile_m = tile / gx, tile_n = tile % gx;
    const int m0 = tile_m * BM, n0 = tile_n * BN, nk = K / BKB;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char *>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char *>(B), 0, 0x7fffffff, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const int voff = lane * 16;
    const long a_plane = (long)(K / BKB) * M * BKB; // bytes of one digit plane
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n)&15) | (((n) >> 4) << 14))
#define DMA(buf, kt)                                                                                                       \
    {                                                                                                                      \
        _Pragma("unroll") for (int i0 = 0; i0 < A_GROUPS; i0 += 4)                                                         \
        {                                                                                                                  \
            const int i = i0 + ws, p = i / (BM / 32), j = i % (BM / 32); /* digit plane, 32-row group */                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(size_t)(lds0 + (buf)*STAGE + p * A_PL + j * 1024), 16, voff, \
                                                     (int)(p * a_plane + ((long)(kt)*M + m0 + 32 * j) * BKB), 0, 0);         \
        }                                                                                                                  \
        _Pragma("unroll") for (int j0 = 0; j0 < B_GROUPS; j0 += 4)                                                         \
        {                                                                                                                  \
            const int j = j0 + ws; /* 32-row groups of B */                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lds0 + (buf)*STAGE + 3 * A_PL + j * 1024), 16, voff, \
                                                     (int)(((long)(kt)*N + n0 + 32 * j) * BKB), 0, 0);                       \
        }                                                                                                                  \
    }
#ifndef PROBE_SPLIT
#define PROBE_SPLIT 0 // 1: the DMA of a tile is issued by BOTH groups (group 0: the first two digit planes of A, group 1: the third + B)
#endif
    constexpr int A0 = 2 * (BM / 32), G0_PER_WAVE = A0 / 4, G1_PER_WAVE = (A_GROUPS - A0 + B_GROUPS) / 4;
    static_assert(A0 % 4 == 0 && (A_GROUPS - A0) % 4 == 0, "split dealt evenly");
#define DMA_PART(buf, kt, FIRST, LAST, WITHB)                                                                              \
    {                                                                                                                      \
        _Pragma("unroll") for (int i0 = FIRST; i0 < LAST; i0 += 4)                                                         \
        {                                                                                                                  \
            const int i = i0 + ws, p = i / (BM / 32), j = i % (BM / 32);                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(size_t)(lds0 + (buf)*STAGE + p * A_PL + j * 1024), 16, voff, \
                                                     (int)(p * a_plane + ((long)(kt)*M + m0 + 32 * j) * BKB), 0, 0);         \
        }                                                                                                                  \
        if (WITHB)                                                                                                         \
        {                                                                                                                  \
            _Pragma("unroll") for (int j0 = 0; j0 < B_GROUPS; j0 += 4)                                                     \
            {                                                                                                              \
                const int j = j0 + ws;                                                                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lds0 + (buf)*STAGE + 3 * A_PL + j * 1024), 16, voff, \
                                                         (int)(((long)(kt)*N + n0 + 32 * j) * BKB), 0, 0);                   \
            }                                                                                                              \
        }                                                                                                                  \
    }
#define WAIT_BATCHES(nb, per)                                                                                              \
    {                                                                                                                      \
        if ((nb) >= 6) WAIT_VM(6 * (per));                                                                                 \
        else if ((nb) == 5) WAIT_VM(5 * (per));                                                                            \
        else if ((nb) == 4) WAIT_VM(4 * (per));                                                                            \
        else if ((nb) == 3) WAIT_VM(3 * (per));                                                                            \
        else if ((nb) == 2) WAIT_VM(2 * (per));                                                                            \
        else if ((nb) == 1) WAIT_VM(per);                                                                                  \
        else WAIT_VM(0);                                                                                                   \
    }
    intx16 acc[3][2][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[d][mi][ni][r] = 0;
    // eight 64 x 64 wave tiles: 4 x 2 over a 256 x 128 block, 2 x 4 over a 128 x 256 one; the two waves of a SIMD are in different groups
    const int wm = BM == 256 ? ws : (ws & 1), wn = BM == 256 ? grp : (ws >> 1) + 2 * grp;
    const int fragA = (wm * 64 + lr) * BKB + lh * 16, fragB = 3 * A_PL + (wn * 64 + lr) * BKB + lh * 16;
    intx4 fa[3][2], fb[2];
#define LD(off) (*reinterpret_cast<const intx4 *>(smem + (off)))
#define BARRIER()                                  \
    {                                              \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    }
#ifndef PROBE_LOCKSTEP
#define PROBE_LOCKSTEP 0 // 1: all eight waves in one phase, one barrier per trip (gemm_planes.h's schedule) instead of ping-pong
#endif
#ifndef PROBE_NOWAIT
#define PROBE_NOWAIT 0 // 1: the staging is issued but never waited for inside the loop (is it the waiting or the issuing that costs?)
#endif
#ifndef PROBE_NOFRAG
#define PROBE_NOFRAG 0 // 1: fragments are read once, before the loop (does the staging compete with the fragment reads for the LDS port?)
#endif
#ifndef PROBE_NODMA
#define PROBE_NODMA 0 // 1: no staging inside the loop (the pace of fragment reads + matrix instructions + barriers)
#endif
#if PROBE_SPLIT
    {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
        {
            if (grp == 0)
                DMA_PART(t, t, 0, A0, false)
            else
                DMA_PART(t, t, A0, A_GROUPS, true)
        }
        if (grp == 0)
            WAIT_VM((STAGES - 2) * G0_PER_WAVE); // tile 0
        else
            WAIT_VM((STAGES - 3) * G1_PER_WAVE); // tiles 0 and 1: group 0 reads tile 1 before group 1's first wait
        BARRIER()
        if (grp == 1)
            BARRIER()
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt)
        {
            const int nxt = cur == 0 ? STAGES - 1 : cur - 1;
            if (kt + STAGES - 1 < nk)
            {
                if (grp == 0)
                    DMA_PART(nxt, kt + STAGES - 1, 0, A0, false)
                else
                    DMA_PART(nxt, kt + STAGES - 1, A0, A_GROUPS, true)
            }
            const int bo = cur * STAGE;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                fb[ni] = LD(bo + fragB + ni * 32 * BKB);
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    fa[d][mi] = LD(bo + d * A_PL + fragA + mi * 32 * BKB);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            BARRIER()
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[d][mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[d][mi], fb[ni], acc[d][mi][ni], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            // issued so far: tiles <= min(nk - 1, kt + STAGES - 1).  group 0 needs its share of tile kt + 1, group 1 its share of tile kt + 2
            const int last = min(nk - 1, kt + STAGES - 1);
            if (grp == 0)
            {
                const int nb = max(0, last - (kt + 1));
                WAIT_BATCHES(nb, G0_PER_WAVE)
            }
            else
            {
                const int nb = max(0, last - (kt + 2));
                WAIT_BATCHES(nb, G1_PER_WAVE)
            }
            BARRIER()
            cur = cur + 1 == STAGES ? 0 : cur + 1;
        }
        if (grp == 0)
            BARRIER()
    }
#else
    if (grp == 0)
    {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            DMA(t, t)
        WAIT_VM((STAGES - 2) * DMA_PER_WAVE);
    }
    BARRIER()
    if (!PROBE_LOCKSTEP && grp == 1)
        BARRIER()
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt)
    {
        const int nxt = cur == 0 ? STAGES - 1 : cur - 1; // stage of tile kt + STAGES - 1 = stage of tile kt - 1
        if (!PROBE_NODMA && grp == 0 && kt + STAGES - 1 < nk)
            DMA(nxt, kt + STAGES - 1)
        const int bo = cur * STAGE;
        if (!PROBE_NOFRAG || kt == 0)
        {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                fb[ni] = LD(bo + fragB + ni * 32 * BKB);
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    fa[d][mi] = LD(bo + d * A_PL + fragA + mi * 32 * BKB);
        }
        if (!PROBE_LOCKSTEP)
        {
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            BARRIER()
            __builtin_amdgcn_s_setprio(1);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[d][mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[d][mi], fb[ni], acc[d][mi][ni], 0, 0, 0);
        if (!PROBE_LOCKSTEP)
            __builtin_amdgcn_s_setprio(0);
        if (!PROBE_NODMA && !PROBE_NOWAIT && grp == 0 && kt + 1 < nk)
        {
            // tile kt + 1 has landed; up to STAGES - 2 newer batches stay in flight
            const int newer = min(STAGES - 2, nk - 2 - kt);
            if (newer >= STAGES - 2)
                WAIT_VM((STAGES - 2) * DMA_PER_WAVE);
            else if (newer == 5)
                WAIT_VM(5 * DMA_PER_WAVE);
            else if (newer == 4)
                WAIT_VM(4 * DMA_PER_WAVE);
            else if (newer == 3)
                WAIT_VM(3 * DMA_PER_WAVE);
            else if (newer == 2)
                WAIT_VM(2 * DMA_PER_WAVE);
            else if (newer == 1)
                WAIT_VM(DMA_PER_WAVE);
            else
                WAIT_VM(0);
        }
        BARRIER()
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
    if (!PROBE_LOCKSTEP && grp == 0)
        BARRIER()
#endif
    int fold = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    fold ^= acc[d][mi][ni][r];
    out[(size_t)blockIdx.x * 512 + tid] = fold;
}

int main()
{
    const int M = 256 * 324, N = 4096, K = 1024; // one target of the W_ih launch at 32 lanes
    signed char *A, *B;
    int *out;
    const size_t a_bytes = (size_t)3 * M * K, b_bytes = (size_t)N * K;
    const int tiles = (M / BM) * (N / BN);
    CHECK(hipMalloc(&A, a_bytes));
    CHECK(hipMalloc(&B, b_bytes));
    CHECK(hipMalloc(&out, (size_t)tiles * 512 * sizeof(int)));
    CHECK(hipMemset(A, 1, a_bytes));
    CHECK(hipMemset(B, 1, b_bytes));
    const size_t lds = (size_t)STAGES * STAGE;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8_loop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(i8_loop_kernel, dim3(tiles), dim3(512), lds, 0, A, B, out, M, N, K);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_tile_us = ms * 1e3 / (tiles / 256.0), ops = 2.0 * 3 * M * (double)N * K;
        printf("i8 loop probe: %d tiles of %d x %d x %d, %d stages of %d KB, in %.3f ms: %.2f us per tile and CU = %.3f us per 32-k trip (768 matrix cycles "
               "per SIMD = 0.41 us at 1.87 GHz), %.1f TB/s into LDS; %.0f int8 TOPS on the three digit products = %.0f fp32-equivalent TFLOP/s "
               "(main loop of the fp16-plane W_ih kernel: 696 GFLOP per target in ~0.95 ms = ~730)\n",
               tiles, BM, BN, K, STAGES, STAGE / 1024, ms, per_tile_us, per_tile_us / (K / 32), (double)tiles * (K / 32) * STAGE / (ms * 1e-3) / 1e12,
               ops / (ms * 1e-3) / 1e12, ops / 3 / (ms * 1e-3) / 1e12);
    }
    int h = 0;
    CHECK(hipMemcpy(&h, out, sizeof h, hipMemcpyDeviceToHost));
    printf("(lane 0 fold %d)\n", h);
    return 0;
}
