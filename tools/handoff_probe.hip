// handoff_probe.hip -- measures cross-workgroup hand-off latency on MI355X for different
// store/load cache-policy flavours, same-XCD vs cross-XCD, idle chip vs all CUs busy.
// Decides the protocol of the persistent LSTM kernel.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { PLAIN = 0, SC0 = 1, SC1 = 2, SC0SC1 = 3, NT = 4 };
static const char *kNames[] = {"plain", "sc0", "sc1", "sc0sc1", "nt"};

template <int ST> __device__ __forceinline__ void st32(unsigned *p, unsigned v)
{
    if (ST == PLAIN) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (ST == SC0) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (ST == SC1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == SC0SC1) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == NT) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ unsigned ld32(unsigned *p)
{
    unsigned v;
    if (LD == PLAIN) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == SC0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == SC1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == SC0SC1) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == NT) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xF;
}

struct Ctl
{
    unsigned count[8];      // blocks seen per XCD
    unsigned arrived;       // grid barrier
    unsigned pad[7];
};

// Every block registers on its XCD, grid barrier, then pairs ping-pong.
// mode_pairing: 0 = one same-XCD pair on XCD 0 (idle chip), 1 = one cross-XCD pair (XCD0 idx0 <-> XCD1 idx0),
//               2 = all blocks paired within their XCD (idx 2k <-> 2k+1), 3 = all paired across XCDs (xcc x <-> x^1, same idx)
template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(Ctl *ctl, unsigned *flags, int iters, int pairing, unsigned long long *result,
                                                unsigned *placement)
{
    const unsigned xcc = xcc_id();
    __shared__ unsigned s_idx;
    if (threadIdx.x == 0)
    {
        s_idx = atomicAdd(&ctl->count[xcc], 1u);
        __threadfence();
        atomicAdd(&ctl->arrived, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const unsigned idx = s_idx;
    if (threadIdx.x == 0) placement[blockIdx.x] = xcc * 1000 + idx;
    if (threadIdx.x != 0) return;
    // decide role: slot = index of the flag pair, role 0 = initiator (A), 1 = responder (B), -1 = idle
    int role = -1, slot = 0;
    if (pairing == 0) { if (xcc == 0 && idx == 0) role = 0; if (xcc == 0 && idx == 1) role = 1; }
    if (pairing == 1) { if (xcc == 0 && idx == 0) role = 0; if (xcc == 1 && idx == 0) role = 1; }
    if (pairing == 2) { role = idx & 1; slot = xcc * 64 + (idx >> 1); }
    if (pairing == 3) { role = xcc & 1; slot = (xcc >> 1) * 64 + idx; }
    if (role < 0) return;
    unsigned *ab = flags + slot * 64, *ba = flags + slot * 64 + 32; // 128-byte separated lines
    unsigned long long t0 = wall_clock64();
    unsigned fails = 0;
    for (int i = 1; i <= iters; ++i)
    {
        if (role == 0)
        {
            st32<ST>(ab, (unsigned)i);
            unsigned spins = 0;
            while (ld32<LD>(ba) != (unsigned)i) { if (++spins > (1u << 20)) { fails++; break; } }
        }
        else
        {
            unsigned spins = 0;
            while (ld32<LD>(ab) != (unsigned)i) { if (++spins > (1u << 20)) { fails++; break; } }
            st32<ST>(ba, (unsigned)i);
        }
        if (fails) break;
    }
    unsigned long long t1 = wall_clock64();
    if (role == 0) { result[slot * 2] = t1 - t0; result[slot * 2 + 1] = fails; }
}

template <int ST, int LD> void run(int pairing, int iters, double clk_mhz)
{
    Ctl *ctl; unsigned *flags, *placement; unsigned long long *result;
    CHECK(hipMalloc(&ctl, sizeof(Ctl))); CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
    CHECK(hipMalloc(&flags, 1024 * 64 * 4)); CHECK(hipMemset(flags, 0, 1024 * 64 * 4));
    CHECK(hipMalloc(&placement, 256 * 4));
    CHECK(hipMalloc(&result, 1024 * 16)); CHECK(hipMemset(result, 0xFF, 1024 * 16));
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(256), dim3(64), 0, 0, ctl, flags, iters, pairing, result, placement);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> r(2048);
    CHECK(hipMemcpy(r.data(), result, 2048 * 8, hipMemcpyDeviceToHost));
    Ctl h; CHECK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
    double sum = 0, mx = 0; int n = 0, nf = 0;
    for (int s = 0; s < 1024; ++s)
        if (r[2 * s] != ~0ull) { double us = (double)r[2 * s] / clk_mhz / iters; sum += us; mx = us > mx ? us : mx; n++; nf += r[2 * s + 1] != 0; }
    static const char *pn[] = {"1 pair same-XCD idle", "1 pair cross-XCD idle", "128 pairs same-XCD", "128 pairs cross-XCD"};
    printf("store=%-7s load=%-7s %-22s pairs=%3d roundtrip avg %.3f us max %.3f us (one-way ~%.3f)  FAILED=%d   xcd counts %u %u %u %u %u %u %u %u\n",
           kNames[ST], kNames[LD], pn[pairing], n, n ? sum / n : -1, mx, n ? sum / n / 2 : -1, nf,
           h.count[0], h.count[1], h.count[2], h.count[3], h.count[4], h.count[5], h.count[6], h.count[7]);
    hipFree(ctl); hipFree(flags); hipFree(placement); hipFree(result);
}

template <int ST, int LD> void run_all(int iters, double clk)
{
    for (int p = 0; p < 4; ++p) run<ST, LD>(p, iters, clk);
}

int main()
{
    int rate_khz = 0;
    CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    double clk = rate_khz / 1000.0; // wall_clock64 ticks per us
    printf("wall clock rate %.1f MHz\n", clk);
    const int iters = 2000;
    run_all<SC1, SC1>(iters, clk);
    run_all<SC0SC1, SC0SC1>(iters, clk);
    run_all<PLAIN, SC1>(iters, clk);
    run_all<SC0, SC1>(iters, clk);
    run_all<PLAIN, SC0SC1>(iters, clk);
    run_all<NT, SC1>(iters, clk);
    run_all<SC1, NT>(iters, clk);
    run_all<PLAIN, NT>(iters, clk);
    run_all<SC0, SC0>(iters, clk);
    run_all<PLAIN, PLAIN>(iters, clk);
    return 0;
}
