// tools/lds_pattern_probe.hip -- what a CU's LDS delivers for 1 KiB ds_read_b128 (and 512 B ds_read_b64) reads by the ORDER in which the 64 lanes
// take the sixteen-byte chunks.  lstm_batch8.h's matrix phase reads h in "fragment order": lane l takes chunk l of a 1 KiB k-step (linear), all eight
// waves the same addresses, and reached 91 B/clk/CU in tools/vmem_issue_probe.hip; gemm_planes_pp.h's swizzled reads reach 256 (tools/lds_probe.hip).
// Is the linear order itself the difference?  Patterns (chunk index of lane l inside the 1 KiB piece):
//   0  l                                  linear
//   1  l ^ ((l >> 5) << 3)                lanes 32-63 shifted by 128 B (banks 32..63 when lanes 0-7 sit on 0..31)
//   2  l ^ (((l >> 4) & 1) << 3)          lanes 16-31 / 48-63 shifted by 128 B
//   3  (l & 7) | ((l >> 5) << 3) | (((l >> 3) & 3) << 4)    lanes 0-7 with 32-39 in one 256 B row
//   4  (l & 3) | ((l >> 4) << 2) | (((l >> 2) & 3) << 4)    lanes 0-3, 16-19, 32-35, 48-51 in one 256 B row
//   5  l ^ (((l >> 3) & 1) << 2) ...      (control: a permutation inside 128 B)
// each with all waves on the SAME 16 KiB (as the recurrence) and with a wave's own 16 KiB.  One 512-thread workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_pattern_probe tools/lds_pattern_probe.hip ; run: tools/lds_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void *lds_ptr;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int chunk_of(int pat, int l)
{
    switch (pat)
    {
    case 0: return l;
    case 1: return l ^ ((l >> 5) << 3);
    case 2: return l ^ (((l >> 4) & 1) << 3);
    case 3: return (l & 7) | ((l >> 5) << 3) | (((l >> 3) & 3) << 4);
    case 4: return (l & 3) | ((l >> 4) << 2) | (((l >> 2) & 3) << 4);
    default: return l ^ (((l >> 3) & 1) << 2);
    }
}

#define RD128(off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(s4) : "v"(a), "n"(off))
#define RD64(off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(s2) : "v"(a), "n"(off))

// WIDE: ds_read_b128 (16 x 1 KiB per round and wave) or ds_read_b64 (32 x 512 B)
template <bool WIDE> __global__ __launch_bounds__(512, 1) void probe(int pat, int own, int rounds, unsigned long long *cyc, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8 * 16384 / 4; i += 512)
        reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const unsigned a = lds0 + (own ? w * 16384 : 0) + (WIDE ? chunk_of(pat, l) * 16 : (pat == 0 ? l * 8 : (l ^ ((l >> 5) << 4)) * 8));
    v4u s4 = {0, 0, 0, 0};
    v2u s2 = {0, 0};
    const long long t0 = clock64();
    for (int r = 0; r < rounds; ++r)
    {
        if (WIDE)
        {
            RD128(0); RD128(1024); RD128(2048); RD128(3072); RD128(4096); RD128(5120); RD128(6144); RD128(7168);
            RD128(8192); RD128(9216); RD128(10240); RD128(11264); RD128(12288); RD128(13312); RD128(14336); RD128(15360);
        }
        else
        {
            RD64(0); RD64(512); RD64(1024); RD64(1536); RD64(2048); RD64(2560); RD64(3072); RD64(3584);
            RD64(4096); RD64(4608); RD64(5120); RD64(5632); RD64(6144); RD64(6656); RD64(7168); RD64(7680);
            RD64(8192); RD64(8704); RD64(9216); RD64(9728); RD64(10240); RD64(10752); RD64(11264); RD64(11776);
            RD64(12288); RD64(12800); RD64(13312); RD64(13824); RD64(14336); RD64(14848); RD64(15360); RD64(15872);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    if (l == 0)
        cyc[blockIdx.x * 8 + w] = (unsigned long long)(t1 - t0);
    sink[blockIdx.x * 512 + tid] = s4[0] ^ s4[3] ^ s2[0] ^ s2[1];
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, rounds = 4000;
    unsigned long long *cyc;
    unsigned *sink;
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * cus * 8);
    (void)hipMalloc(&sink, sizeof(unsigned) * cus * 512);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    printf("# tools/lds_pattern_probe: eight waves per CU, 16 KiB per wave and round; bytes per counter tick (clock64) and CU\n");
    for (int wide = 1; wide >= 0; --wide)
        for (int own = 0; own < 2; ++own)
            for (int pat = 0; pat < (wide ? 6 : 2); ++pat)
            {
                for (int rep = 0; rep < 2; ++rep) // (first = warm-up)
                {
                    if (wide)
                        hipLaunchKernelGGL(probe<true>, dim3(cus), dim3(512), 8 * 16384, 0, pat, own, rounds, cyc, sink);
                    else
                        hipLaunchKernelGGL(probe<false>, dim3(cus), dim3(512), 8 * 16384, 0, pat, own, rounds, cyc, sink);
                    (void)hipDeviceSynchronize();
                }
                std::vector<unsigned long long> h(cus * 8);
                (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * cus * 8, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                const double med = (double)h[h.size() / 2] / rounds;
                printf("%s  pattern %d  %s : %8.1f ticks per round  = %6.1f B/tick/CU\n", wide ? "ds_read_b128" : "ds_read_b64 ", pat, own ? "a wave's own 16 KiB " : "all waves same 16 KiB", med,
                       8.0 * 16384 / med);
            }
    return 0;
}
