// tools/mfma16_rate_probe.hip -- what does one v_mfma_f32_16x16x32_f16 cost when TWO waves of a SIMD issue them, and does it matter whether
// the A operand (the weights of csrc/lstm_batch8.h: 128 registers per wave, live for the whole layer) sits in VGPRs or in AGPRs?
// The matrix phase of lstm_batch8_kernel takes ~1,650 cycles for 34 instructions per wave, two waves per SIMD = ~24 cycles per instruction
// where the pipe's rate says 16 (neither LDS latency nor fragment bytes: profiles/r05_lstm_batch8_turns.txt).
// Each wave runs REPS x 32 matrix instructions on four independent accumulators (the kernel has two) with 32 different A fragments
// (weights) and a B fragment that changes every two instructions, and reports cycles per instruction (s_memtime around the loop, the
// slowest wave of the grid).  Variants: A in VGPRs ("v") / AGPRs ("a"); 1 or 2 waves per SIMD (256 / 512 threads, one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma16_rate_probe tools/mfma16_rate_probe.hip && tools/mfma16_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

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

constexpr int REPS = 256, NW = 32; // 32 weight fragments = 128 registers, like the kernel

template <bool AGPR> __device__ __forceinline__ void mfma(floatx4 &acc, const f16x8 &a, const f16x8 &b)
{
    if (AGPR)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <bool AGPR> __global__ __launch_bounds__(512, 2) void probe(const f16x8 *w, float *out, unsigned long long *cycles)
{
    const int tid = threadIdx.x;
    f16x8 W[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i)
        W[i] = w[(i * 64 + (tid & 63))];
    f16x8 b = w[tid & 63];
    floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; ++r)
    {
#pragma unroll
        for (int i = 0; i < NW; ++i)
        {
            mfma<AGPR>(acc[i & 3], W[i], b);
            if (i & 1)
                b[0] = (_Float16)((float)b[0] + 1.0f); // a new B fragment every two instructions (the kernel: one ds_read_b128 per k-step)
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + tid] = s;
    if ((tid & 63) == 0)
        cycles[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
}

template <bool AGPR> static void run(const char *name, int threads, const f16x8 *w, float *out, unsigned long long *cyc)
{
    const int blocks = 256, waves = blocks * threads / 64;
    for (int it = 0; it < 2; ++it)
    {
        hipLaunchKernelGGL(probe<AGPR>, dim3(blocks), dim3(threads), 0, 0, w, out, cyc);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0, mn = ~0ull;
    double sum = 0;
    for (auto c : h)
    {
        mx = c > mx ? c : mx;
        mn = c < mn ? c : mn;
        sum += (double)c;
    }
    const double n = (double)REPS * NW;
    // (s_memtime counts at a constant 100 MHz on this part: the ratio between variants is the result; x shader clock / 100 MHz = shader cycles)
    printf("%-34s %d waves per SIMD: counter ticks per matrix instruction min %.3f avg %.3f max %.3f\n", name, threads / 256, mn / n, sum / waves / n, mx / n);
}

int main()
{
    f16x8 *w;
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&w, (NW + 1) * 64 * sizeof(f16x8)));
    CHECK(hipMemset(w, 0, (NW + 1) * 64 * sizeof(f16x8)));
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int threads = 256; threads <= 512; threads *= 2)
    {
        run<false>("A operand in VGPRs", threads, w, out, cyc);
        run<true>("A operand in AGPRs", threads, w, out, cyc);
        // wall time per instruction and SIMD (events): independent of the counter's rate
        for (int ag = 0; ag < 2; ++ag)
        {
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < 20; ++it)
            {
                if (ag)
                    hipLaunchKernelGGL(probe<true>, dim3(256), dim3(threads), 0, 0, w, out, cyc);
                else
                    hipLaunchKernelGGL(probe<false>, dim3(256), dim3(threads), 0, 0, w, out, cyc);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double per = ms * 1e6 / 20 / ((double)REPS * NW * (threads / 256)); // ns per matrix instruction of one SIMD
            printf("  %s, %d waves per SIMD: %.2f ns per matrix instruction and SIMD (16 cycles at 2.4 GHz = 6.67 ns)\n", ag ? "AGPRs" : "VGPRs", threads / 256, per);
        }
    }
    return 0;
}
