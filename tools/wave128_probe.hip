// tools/wave128_probe.hip -- TIMING probe (results are not checked; run once at the very end of round 3: 1.78 us per 32-k trip, SLOWER
// than gemm_planes_pp.h's 1.28-1.47 -- profiles/r03_wave128_probe.log; 256 VGPRs + 256 AGPRs, one spilled register) of the main loop DESIGN 10.1 proposes for the plane GEMMs after
// tools/i8_loop_probe.hip showed that the LDS port limits them: FOUR waves per 256 x 256 block, one per SIMD, each a 128 x 128 wave
// tile whose 256 accumulator registers live in AGPRs (a wave may have 512 registers at one wave per SIMD), two fp16 planes of A
// against one plane of B (the u8-weight GEMMs), v_mfma_f32_32x32x16_f16.  Fragment bytes per 16-k step and wave: 2 x 4 + 4 = 12 KiB for
// 32 matrix instructions (gemm_planes_pp.h: 10 KiB for 16), i.e. 96 + 48 KB through the LDS port per 2,048 matrix cycles = 70 B/clk
// instead of 101.  No partner wave: the fragments of 16-k step s + 1 are read while the matrix instructions of step s run (two
// fragment sets, software-pipelined by hand through sched_group_barrier-free straight-line code: the compiler keeps the order).
// Staging as in gemm_planes.h (row-major planes, 64-byte rows, XOR-swizzled chunks), three 48 KB stages, one barrier per 32-k trip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/wave128_probe tools/wave128_probe.hip && tools/wave128_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, A_PL = BM * 64, B_PL = BN * 64, STAGE = 2 * A_PL + B_PL, STAGES = 3;
constexpr int DMA_PER_WAVE = (2 * (BM / 16) + BN / 16) / 4; // 48 wave-instructions of 1 KiB per tile, 12 per wave

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

// A: planes [2][M][K] fp16, B: [N][K] fp16 (K contiguous), out: one float per thread
__global__ __launch_bounds__(256, 1) void wave128_kernel(const unsigned short *A, const unsigned short *B, float *out, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int gx = N / BN, tile_m = blockIdx.x / gx, tile_n = blockIdx.x % gx, m0 = tile_m * BM, n0 = tile_n * BN, nk = K / 32;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(B), 0, 0x7fffffff, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const int st_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int voffA = ((lane >> 2) * K) * 2 + st_chunk * 16, voffB = voffA;
    const long a_plane = (long)M * K * 2;
#define DMA(buf, k0)                                                                                                       \
    {                                                                                                                      \
        _Pragma("unroll") for (int i0 = 0; i0 < 2 * (BM / 16); i0 += 4)                                                    \
        {                                                                                                                  \
            const int i = i0 + wave, p = i / (BM / 16), j = i % (BM / 16);                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(size_t)(lds0 + (buf)*STAGE + p * A_PL + j * 1024), 16, voffA, \
                                                     (int)(p * a_plane + ((long)(m0 + 16 * j) * K + (k0)) * 2), 0, 0);       \
        }                                                                                                                  \
        _Pragma("unroll") for (int j0 = 0; j0 < BN / 16; j0 += 4)                                                          \
        {                                                                                                                  \
            const int j = j0 + wave;                                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(size_t)(lds0 + (buf)*STAGE + 2 * A_PL + j * 1024), 16, voffB, \
                                                     (int)(((long)(n0 + 16 * j) * K + (k0)) * 2), 0, 0);                     \
        }                                                                                                                  \
    }
    floatx16 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[mi][ni][r] = 0.f;
    const int sw = (lr >> 2) & 3;
    const int fragA = (wm * 128 + lr) * 64, fragB = 2 * A_PL + (wn * 128 + lr) * 64;
#define LD(off) (*reinterpret_cast<const f16x8 *>(smem + (off)))
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | ((n)&15) | (((n) >> 4) << 14))
    // fragments of one 16-k step: fa[plane][mi], fb[ni]
#define LOAD_FRAGS(FA, FB, bo, kk)                                                                                         \
    {                                                                                                                      \
        const int co = (((kk)*2 + lh) ^ sw) * 16;                                                                          \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) FB[ni] = LD((bo) + fragB + ni * 32 * 64 + co);                    \
        _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                      \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) FA[p][mi] = LD((bo) + fragA + p * A_PL + mi * 32 * 64 + co);  \
    }
#define MMA(FA, FB)                                                                                                        \
    _Pragma("unroll") for (int p = 1; p >= 0; --p)                                                                         \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                   \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                               \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[p][mi], FB[ni], acc[mi][ni], 0, 0, 0);
    f16x8 fa0[2][4], fb0[4], fa1[2][4], fb1[4];
    DMA(0, 0)
    DMA(1, 32)
    WAIT_VM(DMA_PER_WAVE);
    __builtin_amdgcn_s_barrier();
    LOAD_FRAGS(fa0, fb0, 0, 0)
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt)
    {
        const int nxt = cur == 0 ? 2 : cur - 1, bo = cur * STAGE;
        if (kt + 2 < nk)
            DMA(nxt, (kt + 2) * 32)
        LOAD_FRAGS(fa1, fb1, bo, 1) // second half of this tile, read under the first half's matrix instructions
        MMA(fa0, fb0)
        if (kt + 1 < nk)
        {
            if (kt + 2 < nk)
                WAIT_VM(DMA_PER_WAVE); // tile kt + 1 has landed, kt + 2 may be in flight
            else
                WAIT_VM(0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // this wave's reads of stage `cur` are done (lgkmcnt(0)) ...
        __builtin_amdgcn_s_barrier();       // ... and every wave's: the stage may be refilled two trips from now; tile kt + 1 is visible
        cur = cur == 2 ? 0 : cur + 1;
        if (kt + 1 < nk)
            LOAD_FRAGS(fa0, fb0, cur * STAGE, 0) // first half of the next tile, under the second half's matrix instructions
        MMA(fa1, fb1)
    }
    float fold = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                fold += acc[mi][ni][r];
    out[(size_t)blockIdx.x * 256 + tid] = fold;
}

int main()
{
    const int M = 256 * 324, N = 4096, K = 1024; // one target of the W_ih launch at 32 lanes
    unsigned short *A, *B;
    float *out;
    const int tiles = (M / BM) * (N / BN);
    CHECK(hipMalloc(&A, (size_t)2 * M * K * 2));
    CHECK(hipMalloc(&B, (size_t)N * K * 2));
    CHECK(hipMalloc(&out, (size_t)tiles * 256 * sizeof(float)));
    CHECK(hipMemset(A, 0, (size_t)2 * M * K * 2));
    CHECK(hipMemset(B, 0, (size_t)N * K * 2));
    const size_t lds = (size_t)STAGES * STAGE;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wave128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(wave128_kernel, dim3(tiles), dim3(256), lds, 0, A, B, out, M, N, K);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_tile_us = ms * 1e3 / (tiles / 256.0);
        printf("wave128 probe: %d tiles of 256 x 256 x %d in %.3f ms: %.2f us per tile and CU = %.3f us per 32-k trip (2,048 matrix cycles per SIMD "
               "= 1.10 us at 1.87 GHz; gemm_planes_pp.h: ~1.28-1.47 us)\n",
               tiles, K, ms, per_tile_us, per_tile_us / (K / 32));
    }
    return 0;
}
