// tools/vmem_issue_probe.hip -- what does a vector-memory instruction cost on the CU's one address path when all eight waves of a
// workgroup issue them?  (End of round 5; result: profiles/r05_vmem_issue_probe.log.  Its question is DESIGN 4.6 / 10.1's: the
// batched recurrence's turn is bounded by ~51 vector-memory instructions per turn -- 32 twelve-byte polls of L2-resident granules with sc1,
// 8 sixteen-byte row requests from HBM, 8 + 2 + 1 stores -- and by what returning loads share with LDS fragment reads.)
// Every workgroup (one per CU, 512 threads) runs REPS rounds of:  each wave issues PER_WAVE loads of the chosen width from a 32 KB
// L2-resident region (lane l of wave w, load i: granule i * 512 + tid, the kernel's own poll map), waits for them (vmcnt(0)), one
// s_barrier.  Reported: shader cycles per round for the slowest wave = issue + latency + the serialisation of the eight waves, and the
// same with the loads of only HALF the waves (twice as many each), with 16-byte loads, and with an LDS fragment-read storm (each wave
// reads 16 KB of LDS per round, as the matrix phase does) running beside the loads.
//   hipcc --offload-arch=gfx950 -O3 -o tools/vmem_issue_probe tools/vmem_issue_probe.hip && tools/vmem_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                  \
    do                                                                            \
    {                                                                             \
        hipError_t _e = (e);                                                      \
        if (_e != hipSuccess)                                                     \
        {                                                                         \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
constexpr int REPS = 2000;

// WIDTH: 3 = buffer_load_dwordx3 (12 of a granule's 16 bytes), 4 = dwordx4; HALF: only waves 0-3 load (8 each); LDS: fragment reads beside
template <int WIDTH, bool HALF, bool LDS> __global__ __launch_bounds__(512, 2) void probe(const unsigned *region, unsigned *out, unsigned long long *cycles)
{
    __shared__ __attribute__((aligned(16))) unsigned char h[16 * 1024];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(region) + (size_t)blockIdx.x * 8192, 0, 32768, 0x00020000);
    for (int i = tid; i < 4096; i += 512)
        reinterpret_cast<unsigned *>(h)[i] = i;
    __syncthreads();
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; ++r)
    {
        const int per = HALF ? 8 : 4;
        if (!HALF || w < 4)
        {
#pragma unroll
            for (int i = 0; i < per; ++i)
            {
                const int g = HALF ? (i * 256 + (w * 64 + l)) : (i * 512 + tid); // 2048 granules of 16 bytes
                if (WIDTH == 3)
                {
                    const v3u32 v = __builtin_amdgcn_raw_buffer_load_b96(rs, g * 16, 0, 16);
                    acc += v[0] + v[1] + v[2];
                }
                else
                {
                    const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rs, g * 16, 0, 16);
                    acc += v[0] + v[1] + v[2] + v[3];
                }
            }
        }
        if (LDS)
        {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
            {
                const uint4 f = *reinterpret_cast<const uint4 *>(h + ks * 1024 + l * 16);
                acc += f.x ^ f.w;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = acc;
    if (l == 0)
        cycles[blockIdx.x * 8 + w] = t1 - t0;
}

template <int WIDTH, bool HALF, bool LDS> static void run(const char *name, const unsigned *region, unsigned *out, unsigned long long *cyc)
{
    for (int it = 0; it < 2; ++it)
    {
        hipLaunchKernelGGL((probe<WIDTH, HALF, LDS>), dim3(256), dim3(512), 0, 0, region, out, cyc);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(256 * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    double sum = 0;
    for (auto c : h)
    {
        mx = c > mx ? c : mx;
        sum += (double)c;
    }
    printf("%-64s cycles per round: avg %.0f, slowest wave %.0f  (32 load instructions per workgroup and round)\n", name, sum / h.size() / REPS, (double)mx / REPS);
}

int main()
{
    unsigned *region, *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&region, (size_t)256 * 32768));
    CHECK(hipMemset(region, 1, (size_t)256 * 32768));
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(unsigned)));
    CHECK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
    run<3, false, false>("12-byte sc1 loads, 4 per wave, all eight waves", region, out, cyc);
    run<4, false, false>("16-byte sc1 loads, 4 per wave, all eight waves", region, out, cyc);
    run<3, true, false>("12-byte sc1 loads, 8 per wave, waves 0-3 only", region, out, cyc);
    run<3, false, true>("12-byte sc1 loads, 4 per wave + 16 KB of LDS fragment reads per wave", region, out, cyc);
    run<4, false, true>("16-byte sc1 loads, 4 per wave + 16 KB of LDS fragment reads per wave", region, out, cyc);
    return 0;
}
