// tools/dma_probe.hip -- L2 -> LDS bandwidth of `buffer_load_dwordx4 ... lds` by the shape of a wave-instruction:
//   half:  16 rows x 64 B  (what gemm_planes.h issues: one 32-k slice of a row = half a 128-byte line)
//   full:   8 rows x 128 B (whole lines)
// The source is a matrix of `rows` x 2 KiB rows; every workgroup sweeps its own 256-row band K slice by K slice, like
// the GEMM's A operand; bands are re-read `reps` times so that the data comes from the L2 / MALL, not HBM.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/dma_probe tools/dma_probe.hip ; run: tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr;

template <int FULL> __global__ __launch_bounds__(1024) void probe(const unsigned short *src, int row_bytes, int reps, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(src), 0, 0x7fffffff, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const long band = (long)blockIdx.x * 256 * row_bytes;
    // per stage: 256 rows x 128 B = 32 KB (two 32-k slices of one plane, or one slice of two planes)
    const int voff = FULL ? (lane >> 3) * row_bytes + (lane & 7) * 16 : (lane >> 2) * row_bytes + (lane & 3) * 16;
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r)
        for (int k0 = 0; k0 < row_bytes; k0 += 128)
        {
            const int buf = ((k0 >> 7) & 1) * 32768;
            if (FULL)
            {
                // 32 groups of 8 rows; 16 waves -> 2 instructions per wave
#pragma unroll
                for (int i = 0; i < 2; ++i)
                {
                    const int g = wave + 16 * i;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(size_t)(lds0 + buf + g * 1024), 16, voff,
                                                             (int)(band + (long)g * 8 * row_bytes + k0), 0, 0);
                }
            }
            else
            {
                // 2 slices x 16 groups of 16 rows; 16 waves -> 2 instructions per wave
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(size_t)(lds0 + buf + i * 16384 + wave * 1024), 16, voff,
                                                             (int)(band + (long)wave * 16 * row_bytes + k0 + 64 * i), 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70 | 2); // one stage in flight
            acc += smem[(lane * 16 + k0) & 65535];
        }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (acc == 0x12345678u)
        *sink = acc;
}

// ---- the alternative to LDS-DMA: the same 16 rows x 64 B per wave-instruction as plain buffer_load_dwordx4 into registers,
// then ds_write_b128 -- DEPTH loads in flight per wave.  If the DMA path moves one 16-byte lane per clock and CU (7.5 TB/s is
// 14 B per clock and CU), this form is bound by the vector memory path (64 B per clock and CU) instead.
template <int DEPTH> __global__ __launch_bounds__(1024) void probe_reg(const unsigned short *src, int row_bytes, int reps, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(src), 0, 0x7fffffff, 0x00020000);
    const long band = (long)blockIdx.x * 256 * row_bytes;
    const int voff = (lane >> 2) * row_bytes + (lane & 3) * 16;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    unsigned acc = 0;
    const int nk = row_bytes / 64; // 64-byte K slices; a wave fetches its 16 rows of every slice
    for (int r = 0; r < reps; ++r)
    {
        v4u q[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(band + (long)wave * 16 * row_bytes + d * 64), 0);
        for (int k = 0; k < nk; k += DEPTH)
        {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
            {
                const v4u cur = q[d];
                if (k + DEPTH + d < nk)
                    q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(band + (long)wave * 16 * row_bytes + (k + DEPTH + d) * 64), 0);
                *reinterpret_cast<v4u *>(smem + (((k + d) & 1) * 16384) + wave * 1024 + lane * 16) = cur;
            }
            acc += smem[(lane * 16 + k) & 32767];
        }
    }
    if (acc == 0x12345678u)
        *sink = acc;
}

// ---- the same sweep with the source coming from HBM (4 GiB, read once): rows as they lie in a row-major plane (every
// wave-instruction takes 64 bytes out of 16 different rows, 5952 or 2048 bytes apart) against a K-tiled plane, where the 256
// rows x 64 bytes of one (row band, K step) are one contiguous 16 KiB block (a wave-instruction = 1 KiB contiguous)
template <int TILED> __global__ __launch_bounds__(1024) void probe_hbm(const unsigned short *src, int row_bytes, int bands_per_wg, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(src), 0, 0xffffffff, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const int voff = TILED ? lane * 16 : (lane >> 2) * row_bytes + (lane & 3) * 16;
    unsigned acc = 0;
    for (int b = 0; b < bands_per_wg; ++b)
    {
        const long band = ((long)blockIdx.x * bands_per_wg + b) * 256 * row_bytes;
        for (int k0 = 0; k0 < row_bytes; k0 += 64)
        {
            const int buf = ((k0 >> 6) % 3) * 16384;
            // 16 groups of 16 rows, one per wave
            const long off = TILED ? band + (long)(k0 >> 6) * 16384 + wave * 1024 : band + (long)wave * 16 * row_bytes + k0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(size_t)(lds0 + buf + wave * 1024), 16, voff, (int)(off & 0x7fffffff) + (int)0, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0f70 | 2); // two stages in flight
            acc += smem[(lane * 16 + k0) & 32767];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (acc == 0x12345678u)
        *sink = acc;
}

static void hbm_runs(unsigned *sink)
{
    for (int row_bytes : {2048, 5952})
    {
        const size_t total = (size_t)1 << 31; // 2 GiB per launch (the buffer resource addresses 2^31 bytes), source never re-read
        const int bands = (int)(total / ((size_t)256 * row_bytes)), per_wg = bands / 256;
        unsigned short *src;
        hipMalloc(&src, total);
        hipMemset(src, 1, total);
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe_hbm<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe_hbm<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int tiled = 0; tiled < 2; ++tiled)
            for (int it = 0; it < 2; ++it)
            {
                hipEventRecord(e0);
                if (tiled)
                    hipLaunchKernelGGL(probe_hbm<1>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, per_wg, sink);
                else
                    hipLaunchKernelGGL(probe_hbm<0>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, per_wg, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                printf("from HBM, rows of %d B, %s: %.3f ms, %.2f TB/s into LDS\n", row_bytes,
                       tiled ? "K-tiled plane (1 KiB contiguous per wave-instruction)" : "row-major plane (16 rows x 64 B per wave-instruction)", ms,
                       (double)per_wg * 256 * 256 * row_bytes / (ms * 1e-3) / 1e12);
            }
        hipFree(src);
    }
}

int main()
{
    const int row_bytes = 2048, rows = 256 * 256, reps = 8; // 128 MiB source: L2 + MALL resident after the first sweep
    unsigned short *src;
    unsigned *sink;
    hipMalloc(&src, (size_t)rows * row_bytes);
    hipMalloc(&sink, 4);
    hipMemset(src, 1, (size_t)rows * row_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int full = 0; full < 2; ++full)
        for (int it = 0; it < 3; ++it)
        {
            hipEventRecord(e0);
            if (full)
                hipLaunchKernelGGL(probe<1>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, reps, sink);
            else
                hipLaunchKernelGGL(probe<0>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, reps, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s lines: %.3f ms, %.2f TB/s into LDS\n", full ? "full (8 rows x 128 B)" : "half (16 rows x 64 B)", ms,
                   (double)rows * row_bytes * reps / (ms * 1e-3) / 1e12);
        }
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_reg<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_reg<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_reg<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int depth : {2, 4, 8})
        for (int it = 0; it < 3; ++it)
        {
            hipEventRecord(e0);
            if (depth == 2)
                hipLaunchKernelGGL(probe_reg<2>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, reps, sink);
            else if (depth == 4)
                hipLaunchKernelGGL(probe_reg<4>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, reps, sink);
            else
                hipLaunchKernelGGL(probe_reg<8>, dim3(256), dim3(1024), 65536, 0, src, row_bytes, reps, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("registers + ds_write_b128, %d loads in flight per wave: %.3f ms, %.2f TB/s into LDS\n", depth, ms,
                   (double)rows * row_bytes * reps / (ms * 1e-3) / 1e12);
        }
    hbm_runs(sink);
    return 0;
}
