// tools/i8_chain_probe.hip -- what the one-track recurrence's matrix phase costs, piece by piece:
//   hipcc --offload-arch=gfx950 -O3 -o tools/i8_chain_probe tools/i8_chain_probe.hip && tools/i8_chain_probe
// One 512-thread workgroup per CU; waves 0..3 run 16 x v_mfma_i32_16x16x64_i8 on (a) two dependent accumulator chains,
// (b) four chains, (c) eight chains + the 8 A-fragment LDS reads of lstm_persistent_body_i8 with the 4-row / 512-byte pitch
// layout and with a 528-byte pitch.  Shader cycles (s_memtime) per repetition, min over workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(512) void probe(long long *out, int reps)
{
    __shared__ __attribute__((aligned(16))) unsigned char dig[4][528 + 16];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    for (int i = tid; i < 4 * 544; i += 512)
        (&dig[0][0])[i] = (unsigned char)(i * 7);
    __syncthreads();
    v4i32 Wb[8];
    for (int k = 0; k < 8; ++k)
        Wb[k] = v4i32{tid + k, tid * 3 + k, tid ^ k, 5 * k};
    const v4i32 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const int arow = (l & 15) < 3 ? (l & 15) : 3;
    const int pitch = (MODE == 3) ? 528 : 512;
    const unsigned char *ap = &dig[0][0] + arow * pitch + 16 * (l >> 4);
    long long best = 1LL << 60;
    v4i32 acc = {0, 0, 0, 0};
    if (w < 4)
        for (int r = 0; r < reps; ++r)
        {
            const long long t0 = clock64();
            v4i32 Af[8];
            if (MODE >= 2)
            {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    Af[k] = *reinterpret_cast<const v4i32 *>(ap + 64 * k);
            }
            else
            {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    Af[k] = v4i32{l + k + r, l, k, r};
            }
            __builtin_amdgcn_sched_barrier(0);
            v4i32 C[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                C[k] = v4i32{0, 0, 0, 0};
            constexpr int NCH = MODE == 0 ? 1 : MODE == 1 ? 2 : 4; // chains per product
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                C[k % NCH] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[k], Wb[k], C[k % NCH], 0, 0, 0);
                C[4 + k % NCH] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[k], ones, C[4 + k % NCH], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc += C[k];
            asm volatile("" : "+v"(acc));
            const long long t1 = clock64();
            if (t1 - t0 < best)
                best = t1 - t0;
        }
    if (l == 0 && w < 4)
        out[blockIdx.x * 4 + w] = best + (acc[0] == 0x12345 ? 1 : 0);
}

// the structure of lstm_persistent_body_i8's step without the hand-off: all 8 waves write three digit bytes, barrier, waves 0..3
// read 8 fragments + 16 matrix instructions (two chains) at priority PRIO, waves 4..7 sleep SLEEP x 64 cycles
template <int PRIO, int SLEEP, int WR = 1, int RD = 1, int MF = 1> __global__ __launch_bounds__(512) void probe_step(long long *out, int reps)
{
    __shared__ __attribute__((aligned(16))) unsigned char dig[2][4][528];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    for (int i = tid; i < 2 * 4 * 528; i += 512)
        (&dig[0][0][0])[i] = 0;
    __syncthreads();
    v4i32 Wb[8];
    for (int k = 0; k < 8; ++k)
        Wb[k] = v4i32{tid + k, tid * 3 + k, tid ^ k, 5 * k};
    const v4i32 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const int arow = (l & 15) < 3 ? (l & 15) : 3;
    long long best = 1LL << 60, sum = 0;
    v4i32 acc = {0, 0, 0, 0};
    unsigned pay = tid * 2654435761u;
    for (int r = 0; r < reps; ++r)
    {
        unsigned char(*dg)[528] = dig[r & 1];
        if (WR == 1)
        {
            dg[0][tid] = (unsigned char)pay;
            dg[1][tid] = (unsigned char)(pay >> 8);
            dg[2][tid] = (unsigned char)(pay >> 16);
        }
        if (WR == 2)
            reinterpret_cast<unsigned *>(&dg[0][0])[tid] = pay;
        pay = pay * 1664525u + 1013904223u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (w < 4)
        {
            const long long t0 = clock64();
            if (PRIO)
                __builtin_amdgcn_s_setprio(1);
            const unsigned char *ap = &dig[r & 1][arow][16 * (l >> 4)];
            v4i32 Af[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                Af[k] = RD ? *reinterpret_cast<const v4i32 *>(ap + 64 * k) : v4i32{(int)pay + k, l, k, r};
            __builtin_amdgcn_sched_barrier(0);
            v4i32 C = {0, 0, 0, 0}, C1 = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                if (MF)
                {
                    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[k], Wb[k], C, 0, 0, 0);
                    C1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[k], ones, C1, 0, 0, 0);
                }
                else
                    C += Af[k];
            }
            acc += C + C1;
            asm volatile("" : "+v"(acc));
            if (PRIO)
                __builtin_amdgcn_s_setprio(0);
            const long long t1 = clock64();
            best = t1 - t0 < best ? t1 - t0 : best;
            sum += t1 - t0;
            pay += acc[0];
        }
        else
            for (int i = 0; i < SLEEP; ++i)
                __builtin_amdgcn_s_sleep(1);
    }
    if (l == 0 && w < 4)
    {
        out[blockIdx.x * 4 + w] = best + (acc[0] == 0x12345 ? 1 : 0);
        out[1024 + blockIdx.x * 4 + w] = sum / reps;
    }
}
template <int PRIO, int SLEEP, int WR = 1, int RD = 1, int MF = 1> void run_step(long long *d, const char *name)
{
    std::vector<long long> h(2048);
    for (int it = 0; it < 2; ++it)
    {
        hipLaunchKernelGGL((probe_step<PRIO, SLEEP, WR, RD, MF>), dim3(256), dim3(512), 0, 0, d, 400);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, 2048 * sizeof(long long), hipMemcpyDeviceToHost);
    long long mn = 1LL << 60, mx = 0, av = 0;
    for (int i = 0; i < 1024; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; av += h[1024 + i]; }
    printf("step structure, %s: matrix phase min %lld .. %lld, mean %lld cycles (incl. ~100 of s_memtime)\n", name, mn, mx, av / 1024);
}

int main()
{
    long long *d;
    hipMalloc(&d, 2048 * sizeof(long long));
    std::vector<long long> h(1024);
    const char *names[] = {"1 chain per product (2 in all), fragments in registers", "2 chains per product, fragments in registers",
                           "4 chains per product + 8 LDS fragment reads, 512-byte row pitch", "4 chains per product + 8 LDS fragment reads, 528-byte row pitch"};
    for (int m = 0; m < 4; ++m)
    {
        for (int it = 0; it < 2; ++it)
        {
            if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, d, 200);
            if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, d, 200);
            if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, d, 200);
            if (m == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, d, 200);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, 1024 * sizeof(long long), hipMemcpyDeviceToHost);
        long long mn = 1LL << 60, mx = 0;
        for (auto v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        printf("16 x v_mfma_i32_16x16x64_i8, %s: %lld .. %lld cycles per repetition (incl. ~100 of s_memtime)\n", names[m], mn, mx);
    }
    run_step<1, 9, 1, 1, 1>(d, "528-byte pitch: byte writes, LDS reads, matrix instructions");
    run_step<1, 9, 1, 1, 0>(d, "528-byte pitch: byte writes, LDS reads, NO matrix instructions");
    run_step<1, 9, 1, 0, 1>(d, "528-byte pitch: byte writes, fragments from registers, matrix instructions");
    run_step<1, 9, 0, 1, 1>(d, "528-byte pitch: NO writes, LDS reads, matrix instructions");
    run_step<1, 9, 2, 1, 1>(d, "528-byte pitch: one dword write, LDS reads, matrix instructions");
    run_step<1, 9, 0, 0, 1>(d, "barrier only, fragments from registers, matrix instructions");
    run_step<0, 0>(d, "no priority, helpers do not sleep");
    run_step<1, 0>(d, "s_setprio 1, helpers do not sleep");
    run_step<1, 9>(d, "s_setprio 1, helpers sleep 9 x 64");
    run_step<0, 9>(d, "no priority, helpers sleep 9 x 64");
    return 0;
}
