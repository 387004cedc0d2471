// tools/lds_probe.hip -- what the LDS port of a CU delivers for EXACTLY the fragment-read pattern of gemm_planes_pp.h
// (PP_LOAD: 20 swizzled ds_read_b128 per wave and 32-k unit for one-plane weights), alone and beside the LDS-DMA writes of
// PP_DMA (48 KB per trip, issued by the four waves of group 0).  VERDICT round 3, weak #5: DESIGN 10.1 argued from a port of
// 128 B/clk; MI355X_MICROARCH.md gives ds_read_b128 = 256 B/clk/CU.  One 512-thread workgroup per CU, three 48 KB stages.
//   mode 0  all eight waves read (no DMA)            mode 1  four waves (one per SIMD) read
//   mode 2  group 1 reads, group 0 issues the DMA    mode 3  DMA alone (group 0)
//   mode 4  all eight waves read, group 0 also issues the DMA
//   mode 5  DMA alone, dealt to all eight waves (6 pieces each)
//   mode 6  DMA alone (group 0), pieces of 8 rows x 128 B (whole lines) instead of 16 rows x 64 B
//   mode 7  DMA alone (group 0), pieces of 1 KiB contiguous (K-tiled planes)
//   mode 8  DMA alone, whole-line pieces dealt to all eight waves
// Output: bytes per shader clock and CU (clock64 = s_memtime ticks = shader cycles), and TB/s over the chip from the wall time.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe tools/lds_probe.hip ; run: tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void *lds_ptr;
constexpr int A_PL = 16384, B_PL = 16384, BUF = 49152, STAGES = 3;

#define RD(off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sinkv) : "v"(a), "n"(off))

template <int MODE> __global__ __launch_bounds__(512, 1) void probe(const unsigned short *src, int row_elems, int trips, unsigned long long *cyc, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, ws = wave & 3, lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 2) & 3;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const bool reads = MODE == 0 || MODE == 4 || (MODE == 1 && grp == 0) || (MODE == 2 && grp == 1);
    const bool dma = ((MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7) && grp == 0) || MODE == 5 || MODE == 8;
    constexpr int DEAL = (MODE == 5 || MODE == 8) ? 8 : 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(src), 0, 0x7fffffff, 0x00020000);
    const int st_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int voff = (MODE == 6 || MODE == 8) ? ((lane >> 3) * row_elems) * 2 + (lane & 7) * 16 : MODE == 7 ? lane * 16 : ((lane >> 2) * row_elems) * 2 + st_chunk * 16;
    const long band = (long)blockIdx.x * 256 * row_elems * 2; // this workgroup's 256 rows (re-read from the L2 / MALL)
    // fragment addresses of PP_LOAD: wave tile 128 x 64 at (wm = grp, wn = ws)
    const unsigned fragA = lds0 + (grp * 128 + lr) * 64, fragB = lds0 + 2 * A_PL + (ws * 64 + lr) * 64;
    v4u sinkv = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    int cur = 0;
    for (int t = 0; t < trips; ++t)
    {
        if (dma)
        {
            const int k0 = (t * 32) % row_elems;
#pragma unroll
            for (int i0 = 0; i0 < 48; i0 += DEAL) // 32 A groups (two planes: here two column halves) + 16 B groups of 1 KiB
            {
                const int i = i0 + (DEAL == 8 ? wave : ws);
                const long src_off = (MODE == 6 || MODE == 8) ? ((long)(8 * (i & 31)) * row_elems + ((t * 64) % row_elems) + (i >> 5) * 64) * 2
                                     : MODE == 7            ? ((long)(t % 10) * 48 + i) * 1024
                                                            : ((long)(16 * (i & 15)) * row_elems + k0 + (i >> 4) * 32) * 2;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(size_t)(lds0 + cur * BUF + i * 1024), 16, voff, (int)(band + src_off), 0, 0);
            }
            constexpr int INFL = 2 * 48 / DEAL; // two trips in flight, as the kernel
            __builtin_amdgcn_s_waitcnt(0x0f70 | (INFL & 15) | ((INFL >> 4) << 14));
        }
        if (reads)
        {
#pragma unroll
            for (int kl = 0; kl < 2; ++kl)
            {
                const unsigned co = ((kl * 2 + lh) ^ sw) * 16;
                {
                    const unsigned a = fragB + cur * BUF + co;
                    RD(0); RD(32 * 64);
                }
                {
                    const unsigned a = fragA + cur * BUF + co;
                    RD(A_PL + 0 * 2048); RD(A_PL + 1 * 2048); RD(A_PL + 2 * 2048); RD(A_PL + 3 * 2048);
                    RD(0 * 2048); RD(1 * 2048); RD(2 * 2048); RD(3 * 2048);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0)
        cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (sinkv.x == 0x12345678u && sinkv.y == 0x9abcdef0u)
        *sink = sinkv.z;
}

template <int MODE> static void run(const char *what, const unsigned short *src, int row_elems, int trips, unsigned long long *d_cyc, unsigned *d_sink)
{
    const int blocks = 256;
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * BUF);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 512, STAGES * BUF>>>(src, row_elems, trips / 8, d_cyc, d_sink); // warm-up
    hipEventRecord(e0);
    probe<MODE><<<blocks, 512, STAGES * BUF>>>(src, row_elems, trips, d_cyc, d_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[blocks / 2];
    const int readers = MODE == 0 || MODE == 4 ? 8 : ((MODE == 3 || MODE >= 5) ? 0 : 4);
    const bool dma = MODE >= 2;
    const double rd_bytes = (double)readers * 20 * 1024 * trips, wr_bytes = dma ? 48.0 * 1024 * trips : 0.0;
    printf("%-52s %8.3f ms  clock %.2f GHz  cycles/trip %7.1f  reads %6.1f B/clk/CU  dma writes %5.1f B/clk/CU  total %6.1f B/clk/CU  "
           "(chip: reads %.1f TB/s, dma %.2f TB/s)\n",
           what, ms, med / (ms * 1e6), med / trips, rd_bytes / med, wr_bytes / med, (rd_bytes + wr_bytes) / med, rd_bytes * blocks / (ms * 1e9),
           wr_bytes * blocks / (ms * 1e9));
}

int main()
{
    const int rows = 256 * 256; // 256 rows per workgroup, swept again and again (L2 / MALL resident)
    unsigned short *src;
    unsigned long long *d_cyc;
    unsigned *d_sink;
    hipMalloc(&src, (size_t)rows * 3072 * 2 + 65536);
    hipMemset(src, 0x11, (size_t)rows * 3072 * 2 + 65536);
    hipMalloc(&d_cyc, 256 * 8);
    hipMalloc(&d_sink, 4);
    const int trips = 20000;
    printf("# tools/lds_probe: PP_LOAD's fragment reads (20 swizzled ds_read_b128 = 20 KiB per wave and trip) and PP_DMA's 48 KB per trip\n");
    {
        const int row_elems = 1024;
        run<0>("mode 0: eight waves read", src, row_elems, trips, d_cyc, d_sink);
        run<1>("mode 1: four waves (one per SIMD) read", src, row_elems, trips, d_cyc, d_sink);
        run<3>("mode 3: LDS-DMA alone (group 0, 48 KB per trip)", src, row_elems, trips, d_cyc, d_sink);
        run<2>("mode 2: group 1 reads beside group 0's LDS-DMA", src, row_elems, trips, d_cyc, d_sink);
        run<4>("mode 4: eight waves read, group 0 also issues the DMA", src, row_elems, trips, d_cyc, d_sink);
        run<5>("mode 5: LDS-DMA alone, dealt to all eight waves", src, row_elems, trips, d_cyc, d_sink);
        run<6>("mode 6: LDS-DMA alone (group 0), 8 rows x 128 B pieces", src, row_elems, trips, d_cyc, d_sink);
        run<7>("mode 7: LDS-DMA alone (group 0), 1 KiB contiguous pieces", src, row_elems, trips, d_cyc, d_sink);
        run<8>("mode 8: LDS-DMA alone, 8 rows x 128 B pieces, all eight waves", src, row_elems, trips, d_cyc, d_sink);
    }
    // the row pitch of the source (the planes' K): do the 16 rows of a piece spread over the L2 channels?
    printf("# mode 5 (16 rows x 64 B pieces, all eight waves) by the row pitch of the source\n");
    for (int row_elems : {1024, 1056, 1088, 1152, 1536, 2048, 2080, 2112, 2976, 3008})
    {
        char what[96];
        snprintf(what, sizeof what, "mode 5, row pitch %d B", row_elems * 2);
        run<5>(what, src, row_elems, trips, d_cyc, d_sink);
    }
    return 0;
}
