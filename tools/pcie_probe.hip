// pcie_probe.hip -- what the host link of this box does, to price bench.py's `value_pcie`:
// pinned H2D / D2H / both at once, at the copy sizes the engine uses (21 MB = one stem of a 60 s segment, 170 MB), as one
// copy and as 128 back-to-back copies on one stream, and D2H while a kernel holds every compute unit (the persistent
// track-batched LSTM grid does: if copies were shader blits they would stall behind it).
//   hipcc --offload-arch=gfx950 -O2 -o tools/pcie_probe tools/pcie_probe.hip && tools/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x)                                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            printf("%s: %s\n", #x, hipGetErrorString(e));                                                             \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)
__global__ __launch_bounds__(1024) void hog_kernel(long long cycles, unsigned *sink)
{
    extern __shared__ unsigned lds[]; // sized so that one workgroup fills a CU
    const long long t0 = clock64();
    unsigned v = threadIdx.x;
    while (clock64() - t0 < cycles)
        v = v * 1664525u + 1013904223u;
    lds[threadIdx.x] = v;
    if (v == 0xdeadbeefu)
        sink[0] = lds[(threadIdx.x + 1) & 1023];
}
// an HBM-bound kernel (read + write of a 2 GB buffer, `passes` times): what the engine's streaming kernels look like to the memory system
__global__ __launch_bounds__(256) void hbm_kernel(float4 *buf, size_t n4, int passes)
{
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        {
            float4 v = buf[i];
            v.x += 1.0f;
            buf[i] = v;
        }
}
// device -> pinned host by a FEW workgroups (zero-copy stores over the link): the alternative to the DMA engines
__global__ __launch_bounds__(256) void zero_copy_kernel(const float4 *src, float4 *dst_host, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        dst_host[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t MB = 1 << 20, big = 2048 * MB;
    char *h_a, *h_b, *d_a, *d_b;
    CK(hipHostMalloc((void **)&h_a, big, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_b, big, hipHostMallocDefault));
    CK(hipMalloc((void **)&d_a, big));
    CK(hipMalloc((void **)&d_b, big));
    memset(h_a, 1, big);
    memset(h_b, 2, big);
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned *sink;
    CK(hipMalloc((void **)&sink, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    auto run = [&](const char *label, size_t chunk, int n, bool h2d, bool d2h, bool hog, int nstreams) -> int {
        for (int rep = 0; rep < 3; ++rep)
        {
            CK(hipDeviceSynchronize());
            if (hog) // ~200 ms of every CU held by a 1024-thread, 150 KB-LDS workgroup
                hipLaunchKernelGGL(hog_kernel, dim3(256), dim3(1024), 150 * 1024, s2, 400000000LL, sink);
            const double t0 = now();
            for (int i = 0; i < n; ++i)
            {
                const size_t off = ((size_t)i * chunk) % (big - chunk + 1);
                hipStream_t sa = nstreams > 1 ? ((i & 1) ? s1 : s0) : s0;
                if (h2d)
                    CK(hipMemcpyAsync(d_a + off, h_a + off, chunk, hipMemcpyHostToDevice, sa));
                if (d2h)
                    CK(hipMemcpyAsync(h_b + off, d_b + off, chunk, hipMemcpyDeviceToHost, (h2d && nstreams == 1) ? s1 : sa));
            }
            CK(hipStreamSynchronize(s0));
            CK(hipStreamSynchronize(s1));
            const double dt = now() - t0;
            CK(hipDeviceSynchronize());
            if (rep == 2)
                printf("%-64s %7.1f ms  %6.1f GB/s%s\n", label, dt * 1e3, (double)chunk * n * ((h2d ? 1 : 0) + (d2h ? 1 : 0)) / dt / 1e9,
                       h2d && d2h ? " (both directions summed)" : "");
        }
        return 0;
    };
    const size_t c21 = 21168000, c170 = 8 * c21;
    if (run("H2D pinned, 1 x 170 MB", c170, 1, true, false, false, 1)) return 1;
    if (run("H2D pinned, 8 x 21 MB, one stream", c21, 8, true, false, false, 1)) return 1;
    if (run("D2H pinned, 1 x 170 MB", c170, 1, false, true, false, 1)) return 1;
    if (run("D2H pinned, 8 x 21 MB, one stream", c21, 8, false, true, false, 1)) return 1;
    if (run("D2H pinned, 96 x 21 MB, one stream", c21, 96, false, true, false, 1)) return 1;
    if (run("D2H pinned, 96 x 21 MB, two streams", c21, 96, false, true, false, 2)) return 1;
    if (run("D2H pinned, 12 x 170 MB, one stream", c170, 12, false, true, false, 1)) return 1;
    if (run("H2D + D2H at once, 12 x 170 MB each, two streams", c170, 12, true, true, false, 1)) return 1;
    if (run("D2H pinned, 12 x 170 MB while a kernel holds every CU", c170, 12, false, true, true, 1)) return 1;
    if (run("H2D pinned, 12 x 170 MB while a kernel holds every CU", c170, 12, true, false, true, 1)) return 1;
    // ---- the same while an HBM-bound kernel runs (the engine's copies overlap the other pipeline slot's kernels)
    {
        auto with_hbm = [&](const char *label, int nstreams, int zero_copy_wgs) -> int {
            for (int rep = 0; rep < 2; ++rep)
            {
                CK(hipDeviceSynchronize());
                hipLaunchKernelGGL(hbm_kernel, dim3(2048), dim3(256), 0, s2, reinterpret_cast<float4 *>(d_a), big / 16, 40);
                const double t0 = now();
                const int n = 12;
                for (int i = 0; i < n; ++i)
                {
                    const size_t off = (size_t)i * c170;
                    hipStream_t sa = nstreams > 1 ? ((i & 1) ? s1 : s0) : s0;
                    if (zero_copy_wgs)
                        hipLaunchKernelGGL(zero_copy_kernel, dim3(zero_copy_wgs), dim3(256), 0, sa, reinterpret_cast<const float4 *>(d_b + off),
                                           reinterpret_cast<float4 *>(h_b + off), c170 / 16);
                    else
                        CK(hipMemcpyAsync(h_b + off, d_b + off, c170, hipMemcpyDeviceToHost, sa));
                }
                CK(hipStreamSynchronize(s0));
                CK(hipStreamSynchronize(s1));
                const double dt = now() - t0;
                const hipError_t busy = hipStreamQuery(s2); // still running = the whole transfer overlapped the kernel
                CK(hipDeviceSynchronize());
                (void)hipGetLastError();
                if (rep == 1)
                    printf("%-64s %7.1f ms  %6.1f GB/s%s\n", label, dt * 1e3, (double)c170 * n / dt / 1e9,
                           busy == hipErrorNotReady ? "" : " (the HBM kernel ended first)");
            }
            return 0;
        };
        CK(hipDeviceSynchronize());
        double t0 = now();
        hipLaunchKernelGGL(hbm_kernel, dim3(2048), dim3(256), 0, s2, reinterpret_cast<float4 *>(d_a), big / 16, 40);
        CK(hipDeviceSynchronize());
        double dt = now() - t0;
        printf("%-64s %7.1f ms  %6.1f GB/s of HBM traffic\n", "the HBM-bound kernel alone (40 passes over 2 GB, r+w)", dt * 1e3, 2.0 * big * 40 / dt / 1e9);
        if (with_hbm("D2H pinned, 12 x 170 MB, one stream, beside the HBM kernel", 1, 0)) return 1;
        if (with_hbm("D2H pinned, 12 x 170 MB, two streams, beside the HBM kernel", 2, 0)) return 1;
        if (with_hbm("zero-copy stores by 16 workgroups, beside the HBM kernel", 1, 16)) return 1;
        if (with_hbm("zero-copy stores by 64 workgroups, beside the HBM kernel", 1, 64)) return 1;
        for (int wgs : {8, 32, 256})
        {
            CK(hipDeviceSynchronize());
            t0 = now();
            hipLaunchKernelGGL(zero_copy_kernel, dim3(wgs), dim3(256), 0, s0, reinterpret_cast<const float4 *>(d_b), reinterpret_cast<float4 *>(h_b), big / 16);
            CK(hipDeviceSynchronize());
            dt = now() - t0;
            char lab[96];
            snprintf(lab, sizeof lab, "zero-copy stores by %d workgroups, alone, 2 GB", wgs);
            printf("%-64s %7.1f ms  %6.1f GB/s\n", lab, dt * 1e3, (double)big / dt / 1e9);
        }
    }
    // ---- does the HOST block while it queues many copies behind a running kernel?  (a runtime that runs out of completion
    // signals makes hipMemcpyAsync wait for earlier commands: the caller then cannot queue the next step's kernels)
    for (int ncopies : {16, 64, 128, 256})
    {
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(hog_kernel, dim3(256), dim3(1024), 150 * 1024, s0, 200000000LL, sink); // ~100 ms on the SAME stream
        const double t0 = now();
        for (int i = 0; i < ncopies; ++i)
            CK(hipMemcpyAsync(h_b + ((size_t)i * c21) % (big - c21), d_b + ((size_t)i * c21) % (big - c21), c21, hipMemcpyDeviceToHost, s0));
        const double t_enq = now() - t0;
        CK(hipStreamSynchronize(s0));
        const double t_all = now() - t0;
        char lab[96];
        snprintf(lab, sizeof lab, "queue %d x 21 MB D2H behind a ~100 ms kernel: host time to queue", ncopies);
        printf("%-64s %7.1f ms  (all done after %.1f ms)\n", lab, t_enq * 1e3, t_all * 1e3);
    }
    // pageable host memory (what a caller that does not pin gets)
    {
        std::vector<char> pg(c170 * 2, 3);
        CK(hipDeviceSynchronize());
        double t0 = now();
        CK(hipMemcpy(d_a, pg.data(), c170 * 2, hipMemcpyHostToDevice));
        double dt = now() - t0;
        printf("%-64s %7.1f ms  %6.1f GB/s\n", "H2D pageable, 1 x 340 MB", dt * 1e3, (double)c170 * 2 / dt / 1e9);
        t0 = now();
        CK(hipMemcpy(pg.data(), d_a, c170 * 2, hipMemcpyDeviceToHost));
        dt = now() - t0;
        printf("%-64s %7.1f ms  %6.1f GB/s\n", "D2H pageable, 1 x 340 MB", dt * 1e3, (double)c170 * 2 / dt / 1e9);
    }
    return 0;
}
