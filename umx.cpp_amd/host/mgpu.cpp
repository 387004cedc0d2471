// mgpu.cpp -- one track over the GPUs of a node, exact (include/umx_mgpu.h): C++17 host, RCCL point to point over xGMI.
// Segment s runs on rank s % world, phase by phase; LSTM layer states travel engine to engine (device pointers, the
// engine's own stream), weighted stems travel to rank 0 on a second communicator and stream.  Same schedule as
// umx_split_inference_carry (host/split.cpp), which the CPU tests exercise over gloo.
#include "../../include/umx_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace
{
void seterr(char *err, const std::string &s)
{
    if (err)
        snprintf(err, 256, "%s", s.c_str());
}
#define MG_HIP(expr)                                                                                                 \
    do                                                                                                               \
    {                                                                                                                \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess)                                                                                        \
        {                                                                                                            \
            seterr(err, std::string(#expr) + ": " + hipGetErrorString(e_));                                          \
            return UMX_ERR_HIP;                                                                                      \
        }                                                                                                            \
    } while (0)
#define MG_NCCL(expr)                                                                                                \
    do                                                                                                               \
    {                                                                                                                \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != ncclSuccess)                                                                                       \
        {                                                                                                            \
            seterr(err, std::string(#expr) + ": " + ncclGetErrorString(r_));                                         \
            return UMX_ERR_HIP;                                                                                      \
        }                                                                                                            \
    } while (0)
#define MG_UMX(expr)                                                                                                 \
    do                                                                                                               \
    {                                                                                                                \
        int rc_ = (expr);                                                                                            \
        if (rc_ != UMX_OK)                                                                                           \
        {                                                                                                            \
            seterr(err, std::string(#expr) + ": " + umx_hip_last_error(m->ctx));                                     \
            return rc_;                                                                                              \
        }                                                                                                            \
    } while (0)
} // namespace

struct umx_mgpu
{
    umx_hip_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t state_comm = nullptr, gather_comm = nullptr;
    hipStream_t gather_stream = nullptr;
    float *arena = nullptr; // device scratch of the track driver, grow-only (hipMalloc / hipFree per track would dominate short tracks)
    size_t arena_floats = 0;
};

extern "C" int umx_mgpu_unique_id(char id[UMX_MGPU_ID_BYTES], char *err)
{
    static_assert(2 * sizeof(ncclUniqueId) <= UMX_MGPU_ID_BYTES, "id buffer too small");
    ncclUniqueId a, b;
    MG_NCCL(ncclGetUniqueId(&a));
    MG_NCCL(ncclGetUniqueId(&b));
    memset(id, 0, UMX_MGPU_ID_BYTES);
    memcpy(id, &a, sizeof a);
    memcpy(id + sizeof a, &b, sizeof b);
    return UMX_OK;
}

extern "C" int umx_mgpu_create(umx_mgpu **out, umx_hip_ctx *ctx, int rank, int world, const char id[UMX_MGPU_ID_BYTES], char *err)
{
    if (!out || !ctx || world < 1 || rank < 0 || rank >= world || (world > 1 && !id))
    {
        seterr(err, "umx_mgpu_create: bad argument");
        return UMX_ERR_ARG;
    }
    umx_mgpu *m = new umx_mgpu;
    m->ctx = ctx;
    m->rank = rank;
    m->world = world;
    *out = m;
    MG_HIP(hipStreamCreateWithFlags(&m->gather_stream, hipStreamNonBlocking));
    if (world > 1)
    {
        ncclUniqueId a, b;
        memcpy(&a, id, sizeof a);
        memcpy(&b, id + sizeof a, sizeof b);
        MG_NCCL(ncclCommInitRank(&m->state_comm, world, a, rank));
        MG_NCCL(ncclCommInitRank(&m->gather_comm, world, b, rank));
    }
    return UMX_OK;
}

extern "C" void umx_mgpu_destroy(umx_mgpu *m)
{
    if (!m)
        return;
    (void)hipDeviceSynchronize();
    if (m->state_comm)
        (void)ncclCommDestroy(m->state_comm);
    if (m->gather_comm)
        (void)ncclCommDestroy(m->gather_comm);
    if (m->gather_stream)
        (void)hipStreamDestroy(m->gather_stream);
    if (m->arena)
        (void)hipFree(m->arena);
    delete m;
}

extern "C" int umx_mgpu_separate_track(umx_mgpu *m, const float *audio_host, int length, int shift_offset,
                                       float *const out_host[4], unsigned flags, char *err)
{
    if (!m || !audio_host || length < 1 || shift_offset >= UMX_MAX_SHIFT || (m->rank == 0 && !out_host))
    {
        seterr(err, "umx_mgpu_separate_track: bad argument");
        return UMX_ERR_ARG;
    }
    umx_hip_ctx *ctx = m->ctx;
    const int rank = m->rank, world = m->world;
    const int N = umx_hip_segment_samples(ctx), Hl = umx_hip_hidden(ctx) / 2;
    // shift_inference (umx.cpp:99-150): the track sits `lead` samples into a zero buffer of L2 samples
    const int lead = shift_offset < 0 ? 0 : shift_offset;
    const long long L2ll = shift_offset < 0 ? (long long)length : (long long)length + std::max(UMX_MAX_SHIFT - shift_offset, shift_offset);
    if (L2ll > 0x3fffffff)
    {
        seterr(err, "track too long");
        return UMX_ERR_ARG;
    }
    const int L2 = (int)L2ll;
    const int stride = (int)((1 - 0.25f) * N); // umx.cpp:181
    std::vector<int> offsets;
    for (long long off = 0; off < L2; off += stride)
        offsets.push_back((int)off);
    const int nseg = (int)offsets.size();
    hipStream_t st = (hipStream_t)umx_hip_phase_stream(ctx);
    float *state = umx_hip_stream_state_device(ctx);

    // device buffers from one grow-only arena: a (2,n) input chunk and 4 stems per local segment (kept until the gather
    // has taken them); on rank 0 also the track accumulators, the weight sum and two receive buffers
    size_t need = 0;
    for (int i = rank; i < nseg; i += world)
        need += (size_t)5 * 2 * std::min(N, L2 - offsets[i]) + 5 * 64;
    if (rank == 0)
        need += (size_t)L2 + (size_t)4 * 2 * L2 + (world > 1 ? (size_t)8 * 2 * N : 0) + 16 * 64;
    if (need > m->arena_floats)
    {
        (void)hipDeviceSynchronize();
        if (m->arena)
            (void)hipFree(m->arena);
        m->arena = nullptr;
        m->arena_floats = 0;
        void *p = nullptr;
        if (hipMalloc(&p, (need + need / 8) * sizeof(float)) != hipSuccess)
        {
            seterr(err, "out of device memory");
            return UMX_ERR_HIP;
        }
        m->arena = static_cast<float *>(p);
        m->arena_floats = need + need / 8;
    }
    size_t used = 0;
    auto dalloc = [&](size_t floats) -> float * {
        float *p = m->arena + used;
        used += (floats + 63) / 64 * 64; // 256-byte granules
        return used <= m->arena_floats ? p : nullptr;
    };
    auto cleanup = [&]() { (void)hipDeviceSynchronize(); };
    struct Local
    {
        int seg, off, n;
        float *in, *stems[4];
        hipEvent_t done;
    };
    std::vector<Local> mine;
    for (int i = rank; i < nseg; i += world)
    {
        Local lc;
        lc.seg = i;
        lc.off = offsets[i];
        lc.n = std::min(N, L2 - lc.off); // umx.cpp:217
        lc.in = dalloc((size_t)2 * lc.n);
        for (int t = 0; t < 4; ++t)
            lc.stems[t] = dalloc((size_t)2 * lc.n);
        if (!lc.in || !lc.stems[3] || hipEventCreateWithFlags(&lc.done, hipEventDisableTiming) != hipSuccess)
        {
            cleanup();
            seterr(err, "out of device memory");
            return UMX_ERR_HIP;
        }
        mine.push_back(lc);
    }
    int rc = UMX_OK;
    auto fail = [&](int code) {
        cleanup();
        for (Local &lc : mine)
            (void)hipEventDestroy(lc.done);
        return code;
    };
#define MG_TRY(...)                                                                                                  \
    if ((rc = [&]() -> int { __VA_ARGS__; return UMX_OK; }()) != UMX_OK)                                             \
        return fail(rc);

    // ---- the segments of this rank, in order.  Everything is queued on the engine's phase stream; nothing waits.
    for (Local &lc : mine)
    {
        MG_TRY(
            // the chunk of the (shifted) track: zeros outside [lead, lead + length)
            MG_HIP(hipMemsetAsync(lc.in, 0, sizeof(float) * 2 * (size_t)lc.n, st));
            const long long lo = std::max<long long>(lc.off, lead), hi = std::min<long long>((long long)lc.off + lc.n, (long long)lead + length);
            if (hi > lo)
                MG_HIP(hipMemcpyAsync(lc.in + 2 * (size_t)(lo - lc.off), audio_host + 2 * (size_t)(lo - lead),
                                      sizeof(float) * 2 * (size_t)(hi - lo), hipMemcpyHostToDevice, st));
            MG_UMX(umx_hip_segment_begin_device(ctx, lc.in, lc.n, flags));
            for (int l = 0; l < 3; ++l)
            {
                // layer l's (h, c) of target t: 4 * Hl floats at ((t * 3 + l) * 4) * Hl of the stream state
                if (lc.seg == 0) // umx.cpp:167-171: a track starts from zero state
                    for (int t = 0; t < 4; ++t)
                        MG_HIP(hipMemsetAsync(state + ((size_t)(t * 3 + l) * 4) * Hl, 0, sizeof(float) * 4 * Hl, st));
                else if (world > 1)
                {
                    MG_NCCL(ncclGroupStart());
                    for (int t = 0; t < 4; ++t)
                        MG_NCCL(ncclRecv(state + ((size_t)(t * 3 + l) * 4) * Hl, (size_t)4 * Hl, ncclFloat, (lc.seg - 1) % world, m->state_comm, st));
                    MG_NCCL(ncclGroupEnd());
                }
                MG_UMX(umx_hip_segment_lstm_layer(ctx, l));
                if (world > 1 && lc.seg + 1 < nseg)
                {
                    MG_NCCL(ncclGroupStart());
                    for (int t = 0; t < 4; ++t)
                        MG_NCCL(ncclSend(state + ((size_t)(t * 3 + l) * 4) * Hl, (size_t)4 * Hl, ncclFloat, (lc.seg + 1) % world, m->state_comm, st));
                    MG_NCCL(ncclGroupEnd());
                }
            }
            MG_UMX(umx_hip_segment_end_device(ctx, lc.stems));
            MG_UMX(umx_hip_weight_stems_device(ctx, lc.stems, lc.n, st)); // umx.cpp:246
            MG_HIP(hipEventRecord(lc.done, st));)
    }

    // ---- gather: rank 0 adds the weighted stems in segment order (umx.cpp:234-260), on its own stream
    if (rank != 0)
    {
        for (Local &lc : mine)
        {
            MG_TRY(MG_HIP(hipStreamWaitEvent(m->gather_stream, lc.done, 0)); MG_NCCL(ncclGroupStart());
                   for (int t = 0; t < 4; ++t) MG_NCCL(ncclSend(lc.stems[t], (size_t)2 * lc.n, ncclFloat, 0, m->gather_comm, m->gather_stream));
                   MG_NCCL(ncclGroupEnd());)
        }
        MG_TRY(MG_HIP(hipStreamSynchronize(m->gather_stream)); MG_HIP(hipStreamSynchronize(st)); MG_UMX(umx_hip_sync(ctx));)
        return fail(UMX_OK);
    }
    float *track[4], *sumw = dalloc((size_t)L2), *rbuf[2][4];
    for (int t = 0; t < 4; ++t)
    {
        track[t] = dalloc((size_t)2 * L2);
        for (int b = 0; b < 2; ++b)
            rbuf[b][t] = world > 1 ? dalloc((size_t)2 * N) : nullptr;
    }
    if (!sumw || !track[3] || (world > 1 && !rbuf[1][3]))
    {
        seterr(err, "out of device memory");
        return fail(UMX_ERR_HIP);
    }
    hipStream_t gs = m->gather_stream;
    MG_TRY(for (int t = 0; t < 4; ++t) MG_HIP(hipMemsetAsync(track[t], 0, sizeof(float) * 2 * (size_t)L2, gs));
           MG_HIP(hipMemsetAsync(sumw, 0, sizeof(float) * (size_t)L2, gs));)
    size_t mi = 0;
    for (int i = 0; i < nseg; ++i)
    {
        const int off = offsets[i], n = std::min(N, L2 - off);
        if (i % world == 0)
        {
            Local &lc = mine[mi++];
            MG_TRY(MG_HIP(hipStreamWaitEvent(gs, lc.done, 0));
                   MG_UMX(umx_hip_track_accumulate_device(ctx, track, sumw, lc.stems, off, n, gs));)
        }
        else
        {
            float *const *rb = rbuf[i & 1]; // two receive buffers: the add of one overlaps the transfer of the next
            MG_TRY(MG_NCCL(ncclGroupStart());
                   for (int t = 0; t < 4; ++t) MG_NCCL(ncclRecv(rb[t], (size_t)2 * n, ncclFloat, i % world, m->gather_comm, gs));
                   MG_NCCL(ncclGroupEnd()); MG_UMX(umx_hip_track_accumulate_device(ctx, track, sumw, rb, off, n, gs));)
        }
    }
    MG_TRY(MG_UMX(umx_hip_track_normalise_device(ctx, track, sumw, L2, gs)); // umx.cpp:264-273
           for (int t = 0; t < 4; ++t) MG_HIP(hipMemcpyAsync(out_host[t], track[t] + 2 * (size_t)lead, sizeof(float) * 2 * (size_t)length,
                                                             hipMemcpyDeviceToHost, gs)); // umx.cpp:136-147: drop the shift
           MG_HIP(hipStreamSynchronize(gs)); MG_HIP(hipStreamSynchronize(st)); MG_UMX(umx_hip_sync(ctx));)
    return fail(UMX_OK);
#undef MG_TRY
}
