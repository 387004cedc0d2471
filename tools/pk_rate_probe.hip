// tools/pk_rate_probe.hip -- what a packed fp32 VALU instruction costs on gfx950 next to its scalar form: issue cycles per
// instruction and wave for v_mul / v_add / v_fma_f32 against v_pk_mul / v_pk_add / v_pk_fma_f32, 8 independent chains per wave,
// one and four waves per SIMD, one workgroup per CU.  (Before packing the fused Wiener kernel's complex algebra: DESIGN 4.8, 10.)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/pk_rate_probe tools/pk_rate_probe.hip ; run: tools/pk_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE> __global__ void probe(float *out, unsigned long long *cyc, int iters, float seed)
{
    v2 r[8], k = {seed, seed * 0.5f}, c = {0.25f, 0.125f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r[i] = v2{seed + i + threadIdx.x, seed - i};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(k.x));
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i].x) : "v"(k.x), "v"(c.x));
#define P_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
#define P_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
#define P_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(c));
            if (MODE == 0) { REP8(S_MUL) }
            if (MODE == 1) { REP8(S_ADD) }
            if (MODE == 2) { REP8(S_FMA) }
            if (MODE == 3) { REP8(P_MUL) }
            if (MODE == 4) { REP8(P_ADD) }
            if (MODE == 5) { REP8(P_FMA) }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) // every wave: the arbiter favours the oldest wave, so one wave's span says nothing about the SIMD
    {
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = (unsigned long long)t0;
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = (unsigned long long)t1;
    }
}

template <int MODE> static void run(const char *what, int threads, float *out, unsigned long long *cyc)
{
    const int iters = 4096, blocks = 256;
    probe<MODE><<<blocks, threads>>>(out, cyc, 64, 1.0f);
    probe<MODE><<<blocks, threads>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    static unsigned long long c[256 * 32];
    hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    unsigned long long m = ~0ull; // the fastest workgroup; a workgroup's span = its first wave's start to its last wave's end
    for (int b = 0; b < blocks; ++b)
    {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < threads / 64; ++w)
        {
            lo = c[(b * 16 + w) * 2] < lo ? c[(b * 16 + w) * 2] : lo;
            hi = c[(b * 16 + w) * 2 + 1] > hi ? c[(b * 16 + w) * 2 + 1] : hi;
        }
        m = hi - lo < m ? hi - lo : m;
    }
    const double per = (double)m / ((double)iters * 32), waves_per_simd = threads / 256.0;
    printf("%-14s %4d threads (%g wave%s per SIMD): %6.2f cycles per instruction of a wave over the workgroup's span = %5.2f cycles of its SIMD per instruction\n", what,
           threads, waves_per_simd, waves_per_simd > 1 ? "s" : "", per, per / waves_per_simd);
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 32 * 8);
    for (int threads : {256, 1024})
    {
        run<0>("v_mul_f32", threads, out, cyc);
        run<1>("v_add_f32", threads, out, cyc);
        run<2>("v_fma_f32", threads, out, cyc);
        run<3>("v_pk_mul_f32", threads, out, cyc);
        run<4>("v_pk_add_f32", threads, out, cyc);
        run<5>("v_pk_fma_f32", threads, out, cyc);
    }
    return 0;
}
