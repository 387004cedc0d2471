// mgpu.cpp -- one track over the GPUs of a node, exact (include/umx_mgpu.h): C++17 host, RCCL point to point over xGMI.
// world = G target groups x P pipeline stages (host/shard_plan.h).  Rank (g, p) runs the targets t % G == g of the
// segments s % P == p, phase by phase; a layer's LSTM state travels stage to stage between the engines' HBM state
// buffers, target magnitudes travel to the rank that filters the segment, weighted stems travel to rank 0.  Same
// schedule as umx_split_inference_carry / _targets (host/split.cpp), which the CPU tests exercise over gloo.
//
// Streams of one rank, and what each may wait for:
//   main   (the engine's phase stream)  kernels + RECEIVES of LSTM state (communicator ring[colour of the incoming edge])
//   send   sends of LSTM state, each behind an event of the layer that produced it (ring[colour of the outgoing edge])
//   mag    target magnitudes, send or receive, in segment order on every rank (mag_comm)
//   gather weighted stems to rank 0, in segment order on every rank (gather_comm); rank 0 adds them on `acc`
// A send never stands in front of something a peer waits for, so the cross-rank waits are the data dependencies
// (segment s - 1 before s, masks before the filter, stems before the sum): acyclic whether RCCL completes sends eagerly
// or only at the matching receive.
#include "../../include/umx_mgpu.h"
#include "shard_plan.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace
{
void seterr(char *err, const std::string &s)
{
    if (err)
        snprintf(err, 256, "%s", s.c_str());
}
} // namespace

struct umx_mgpu
{
    umx_hip_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    bool loopback = false, by_target = false, dead = false;
    ncclComm_t ring[3] = {nullptr, nullptr, nullptr}, gather_comm = nullptr, mag_comm = nullptr;
    hipStream_t send_stream = nullptr, gather_stream = nullptr, acc_stream = nullptr, mag_stream = nullptr;
    hipEvent_t layer_done[3] = {}, sent[3] = {}, mark = nullptr, mags_sent = nullptr, mags_here = nullptr, recvd[2] = {}, added[2] = {};
    float *arena = nullptr; // device scratch of the track driver, grow-only (hipMalloc / hipFree per track would dominate short tracks)
    size_t arena_floats = 0;
    int *status_dev = nullptr;
    long long stats[5] = {0, 0, 0, 0, 0};
    int reserve_device = 0, reserved_cus = 0; // umx_hip_gate_reserve request of this driver (given back by umx_mgpu_destroy)
    long long timeout_ms = 120000;            // watchdog of every host-side wait of a track (UMX_MGPU_TIMEOUT_MS)
    int debug_stall_hop = -1;                 // testing (UMX_MGPU_DEBUG_STALL=k, loopback): in front of the k-th state hop the engine's stream
                                              // is held by a sleeping host function for twice the deadline -- what a peer that
                                              // does not post its side of a transfer looks like to this rank
    int hop_counter = 0;

    void abort_comms() // a rank that gives up must not leave kernels of its own spinning in RCCL
    {
        for (ncclComm_t *c : {&ring[0], &ring[1], &ring[2], &gather_comm, &mag_comm})
            if (*c)
            {
                (void)ncclCommAbort(*c);
                *c = nullptr;
            }
        dead = true;
    }
};

#define MG_FAIL(code, msg)                                                                                           \
    do                                                                                                               \
    {                                                                                                                \
        seterr(err, msg);                                                                                            \
        return code;                                                                                                 \
    } while (0)
#define MG_HIP(expr)                                                                                                 \
    do                                                                                                               \
    {                                                                                                                \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess)                                                                                        \
            MG_FAIL(UMX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
    } while (0)
#define MG_NCCL(expr)                                                                                                \
    do                                                                                                               \
    {                                                                                                                \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != ncclSuccess)                                                                                       \
            MG_FAIL(UMX_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_));                                \
    } while (0)
#define MG_UMX(expr)                                                                                                 \
    do                                                                                                               \
    {                                                                                                                \
        int rc_ = (expr);                                                                                            \
        if (rc_ != UMX_OK)                                                                                           \
            MG_FAIL(rc_, std::string(#expr) + ": " + umx_hip_last_error(m->ctx));                                    \
    } while (0)

// Every host-side wait of a track goes through here: the streams are POLLED (hipStreamQuery), the communicators are asked
// for asynchronous errors (ncclCommGetAsyncError: a peer that died, a link that failed), and a deadline bounds the whole
// thing (a peer that simply never posts its side of a transfer reports no error at all).  On any of the three this rank
// aborts its communicators -- its own RCCL kernels stop spinning, its streams drain -- and returns an error: no rank waits
// for ever on a peer, with or without a launcher that would tear the job down.
static void debug_stall_fn(void *ms) { std::this_thread::sleep_for(std::chrono::milliseconds(reinterpret_cast<intptr_t>(ms))); }

static int wait_streams(umx_mgpu *m, std::initializer_list<hipStream_t> streams, const char *what, char *err)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin)
    {
        bool all = true;
        for (hipStream_t s : streams)
        {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipErrorNotReady)
                all = false;
            else if (q != hipSuccess)
                MG_FAIL(UMX_ERR_HIP, std::string("hipStreamQuery (") + what + "): " + hipGetErrorString(q));
        }
        (void)hipGetLastError(); // "not ready" is reported as an error
        if (all)
            return UMX_OK;
        for (ncclComm_t c : {m->ring[0], m->ring[1], m->ring[2], m->gather_comm, m->mag_comm})
            if (c)
            {
                ncclResult_t r = ncclSuccess;
                if (ncclCommGetAsyncError(c, &r) != ncclSuccess || (r != ncclSuccess && r != ncclInProgress))
                    MG_FAIL(UMX_ERR_HIP, std::string("RCCL reported an asynchronous error while waiting for ") + what + ": " + ncclGetErrorString(r));
            }
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > m->timeout_ms)
            MG_FAIL(UMX_ERR_TIMEOUT, std::string("watchdog: ") + what + " not finished after " + std::to_string(ms) +
                                         " ms (UMX_MGPU_TIMEOUT_MS): a peer rank is gone or never posted its side of a transfer");
        if (spin < 2000)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

extern "C" int umx_mgpu_unique_id(char id[UMX_MGPU_ID_BYTES], char *err)
{
    static_assert(5 * sizeof(ncclUniqueId) <= UMX_MGPU_ID_BYTES, "id buffer too small");
    memset(id, 0, UMX_MGPU_ID_BYTES);
    for (int i = 0; i < 5; ++i)
    {
        ncclUniqueId u;
        MG_NCCL(ncclGetUniqueId(&u));
        memcpy(id + i * sizeof u, &u, sizeof u);
    }
    return UMX_OK;
}

static int create_impl(umx_mgpu *m, const char *id, char *err)
{
    const int world = m->world, rank = m->rank;
    for (hipStream_t *s : {&m->send_stream, &m->gather_stream, &m->acc_stream, &m->mag_stream})
        MG_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    for (hipEvent_t *e : {&m->layer_done[0], &m->layer_done[1], &m->layer_done[2], &m->sent[0], &m->sent[1], &m->sent[2], &m->mark,
                          &m->mags_sent, &m->mags_here, &m->recvd[0], &m->recvd[1], &m->added[0], &m->added[1]})
        MG_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    MG_HIP(hipMalloc(reinterpret_cast<void **>(&m->status_dev), 64));
    if (world == 1 && !m->loopback)
        return UMX_OK;
    ncclUniqueId ids[5];
    if (world == 1) // loopback: nobody to rendezvous with
        for (ncclUniqueId &u : ids)
            MG_NCCL(ncclGetUniqueId(&u));
    else
        memcpy(ids, id, sizeof ids);
    const umx_plan::Plan pl = umx_plan::make_plan(world, m->by_target);
    // the colours this node's rings use: P = 1 none (loopback: one), even P two, odd P three
    const int ncolours = pl.P == 1 ? (m->loopback ? 1 : 0) : (pl.P % 2 == 0 ? 2 : 3);
    for (int c = 0; c < ncolours; ++c)
        MG_NCCL(ncclCommInitRank(&m->ring[c], world, ids[c], rank));
    MG_NCCL(ncclCommInitRank(&m->gather_comm, world, ids[3], rank));
    if (pl.G > 1 || m->loopback)
        MG_NCCL(ncclCommInitRank(&m->mag_comm, world, ids[4], rank));
    // RCCL's kernels sit on compute units for as long as a transfer waits for its peer: keep them out of the budget
    // of the persistent LSTM grids (which need all their workgroups resident at once)
    const char *e = getenv("UMX_MGPU_RESERVE_CUS");
    MG_HIP(hipGetDevice(&m->reserve_device)); // the engine's device: this thread made it current to create the context
    m->reserved_cus = std::max(0, e ? atoi(e) : 16);
    if (m->reserved_cus > 0 && umx_hip_gate_reserve(m->reserve_device, m->reserved_cus) != UMX_OK)
        m->reserved_cus = 0;
    return UMX_OK;
}

extern "C" int umx_mgpu_create_ex(umx_mgpu **out, umx_hip_ctx *ctx, int rank, int world, const char id[UMX_MGPU_ID_BYTES],
                                  unsigned mgpu_flags, char *err)
{
    if (!out || !ctx || world < 1 || rank < 0 || rank >= world || (world > 1 && !id))
        MG_FAIL(UMX_ERR_ARG, "umx_mgpu_create: bad argument");
    if (umx_hip_lstm_is_batched(ctx) || umx_hip_n_tracks(ctx) != 1)
        MG_FAIL(UMX_ERR_ARG, "umx_mgpu_create: needs a single-track context (its LSTM workgroups share compute units with RCCL's "
                             "kernels; the track-batched kernels need every CU to themselves)");
    if (const char *e = getenv("UMX_MGPU_LOOPBACK"))
        if (atoi(e))
            mgpu_flags |= UMX_MGPU_LOOPBACK;
    umx_mgpu *m = new umx_mgpu;
    m->ctx = ctx;
    m->rank = rank;
    m->world = world;
    m->loopback = world == 1 && (mgpu_flags & UMX_MGPU_LOOPBACK);
    if (const char *e = getenv("UMX_MGPU_TIMEOUT_MS"))
        m->timeout_ms = std::max(1LL, atoll(e));
    if (const char *e = getenv("UMX_MGPU_DEBUG_STALL"))
        m->debug_stall_hop = atoi(e);
    m->by_target = mgpu_flags & UMX_MGPU_BY_TARGET;
    const int rc = create_impl(m, id, err);
    if (rc != UMX_OK)
    {
        umx_mgpu_destroy(m);
        *out = nullptr;
        return rc;
    }
    *out = m;
    return UMX_OK;
}

extern "C" int umx_mgpu_create(umx_mgpu **out, umx_hip_ctx *ctx, int rank, int world, const char id[UMX_MGPU_ID_BYTES], char *err)
{
    return umx_mgpu_create_ex(out, ctx, rank, world, id, 0, err);
}

extern "C" void umx_mgpu_destroy(umx_mgpu *m)
{
    if (!m)
        return;
    if (!m->dead)
        (void)hipDeviceSynchronize();
    for (ncclComm_t c : {m->ring[0], m->ring[1], m->ring[2], m->gather_comm, m->mag_comm})
        if (c)
            (void)ncclCommDestroy(c);
    for (hipStream_t s : {m->send_stream, m->gather_stream, m->acc_stream, m->mag_stream})
        if (s)
            (void)hipStreamDestroy(s);
    for (hipEvent_t e : {m->layer_done[0], m->layer_done[1], m->layer_done[2], m->sent[0], m->sent[1], m->sent[2], m->mark, m->mags_sent,
                         m->mags_here, m->recvd[0], m->recvd[1], m->added[0], m->added[1]})
        if (e)
            (void)hipEventDestroy(e);
    if (m->arena)
        (void)hipFree(m->arena);
    if (m->status_dev)
        (void)hipFree(m->status_dev);
    if (m->reserved_cus > 0) // this driver's own request, on the device it was filed for -- other drivers' requests stay
        (void)umx_hip_gate_reserve(m->reserve_device, -m->reserved_cus);
    delete m;
}

extern "C" int umx_mgpu_stats(const umx_mgpu *m, long long out5[5])
{
    if (!m || !out5)
        return UMX_ERR_ARG;
    memcpy(out5, m->stats, sizeof m->stats);
    return UMX_OK;
}

namespace
{
struct Local // a segment this rank runs
{
    int seg, off, n;
    float *in, *stems[4]; // stems: only where this rank filters the segment
    hipEvent_t done;
};

// One attempt at the whole track; UMX_ERR_TIMEOUT (every rank agrees on it) asks for another one.
int separate_once(umx_mgpu *m, const float *audio_host, int length, int shift_offset, float *const out_host[4], unsigned flags,
                  int *global_status, char *err)
{
    umx_hip_ctx *ctx = m->ctx;
    const int rank = m->rank, world = m->world;
    const bool loop = m->loopback;
    const umx_plan::Plan pl = umx_plan::make_plan(world, m->by_target);
    const int N = umx_hip_segment_samples(ctx), Hl = umx_hip_hidden(ctx) / 2;
    const int my_stage = pl.stage(rank);
    // shift_inference (umx.cpp:99-150): the track sits `lead` samples into a zero buffer of L2 samples
    const int lead = shift_offset < 0 ? 0 : shift_offset;
    const long long L2ll = shift_offset < 0 ? (long long)length : (long long)length + std::max(UMX_MAX_SHIFT - shift_offset, shift_offset);
    if (L2ll > 0x3fffffff)
        MG_FAIL(UMX_ERR_ARG, "track too long");
    const int L2 = (int)L2ll;
    const int stride = (int)((1 - 0.25f) * N); // umx.cpp:181
    std::vector<int> offsets;
    for (long long off = 0; off < L2; off += stride)
        offsets.push_back((int)off);
    const int nseg = (int)offsets.size();
    hipStream_t st = (hipStream_t)umx_hip_phase_stream(ctx), ss = m->send_stream, ms = m->mag_stream, gs = m->gather_stream, as = m->acc_stream;
    float *state = umx_hip_stream_state_device(ctx);
    size_t mag_floats = 0;
    float *mag[4];
    for (int t = 0; t < 4; ++t)
        mag[t] = umx_hip_target_mag_device(ctx, t, &mag_floats);
    // targets of this rank's group; the engine skips the others (their magnitudes arrive from the other groups)
    int mine_t[4], n_mine = 0;
    unsigned eng_flags = flags;
    for (int t = 0; t < 4; ++t)
        if (pl.owns_target(rank, t))
            mine_t[n_mine++] = t;
        else
            eng_flags |= UMX_FLAG_SKIP_TARGET(t);
    const size_t lstate = (size_t)4 * Hl; // one layer of one target: [2 dirs][h, c][Hl]
    auto state_of = [&](int t, int l) { return state + ((size_t)(t * 3 + l) * 4) * Hl; };
    memset(m->stats, 0, 4 * sizeof(long long));

    // ---- device buffers from one grow-only arena
    size_t need = 64;
    for (int i = 0; i < nseg; ++i)
        if (pl.runs_segment(rank, i))
            need += (size_t)(pl.wiener_rank(i) == rank ? 5 : 1) * (2 * (size_t)std::min(N, L2 - offsets[i]) + 64);
    if (rank == 0)
        need += (size_t)L2 + (size_t)4 * 2 * L2 + ((world > 1 || loop) ? (size_t)8 * 2 * N : 0) + 16 * 64;
    if (loop)
        need += 12 * (lstate + 64) + 4 * (mag_floats + 64);
    if (need > m->arena_floats)
    {
        (void)hipDeviceSynchronize();
        if (m->arena)
            (void)hipFree(m->arena);
        m->arena = nullptr;
        m->arena_floats = 0;
        void *p = nullptr;
        if (hipMalloc(&p, (need + need / 8) * sizeof(float)) != hipSuccess)
            MG_FAIL(UMX_ERR_HIP, "out of device memory");
        m->arena = static_cast<float *>(p);
        m->arena_floats = need + need / 8;
    }
    size_t used = 0;
    auto dalloc = [&](size_t floats) -> float * {
        float *p = m->arena + used;
        used += (floats + 63) / 64 * 64; // 256-byte granules
        return used <= m->arena_floats ? p : nullptr;
    };
    std::vector<Local> mine;
    struct EventGuard // the per-segment events die with this attempt, whichever way it ends
    {
        std::vector<Local> &v;
        ~EventGuard()
        {
            for (Local &lc : v)
                if (lc.done)
                    (void)hipEventDestroy(lc.done);
        }
    } guard{mine};
    for (int i = 0; i < nseg; ++i)
    {
        if (!pl.runs_segment(rank, i))
            continue;
        Local lc;
        lc.seg = i;
        lc.off = offsets[i];
        lc.n = std::min(N, L2 - lc.off); // umx.cpp:217
        lc.in = dalloc((size_t)2 * lc.n);
        lc.done = nullptr;
        for (int t = 0; t < 4; ++t)
            lc.stems[t] = pl.wiener_rank(i) == rank ? dalloc((size_t)2 * lc.n) : nullptr;
        if (!lc.in || (pl.wiener_rank(i) == rank && !lc.stems[3]))
            MG_FAIL(UMX_ERR_HIP, "out of device memory");
        MG_HIP(hipEventCreateWithFlags(&lc.done, hipEventDisableTiming));
        mine.push_back(lc);
    }
    float *stash[3][4] = {}, *mag_stash[4] = {};
    if (loop)
    {
        for (int l = 0; l < 3; ++l)
            for (int t = 0; t < 4; ++t)
                if (!(stash[l][t] = dalloc(lstate)))
                    MG_FAIL(UMX_ERR_HIP, "out of device memory");
        for (int t = 0; t < 4; ++t)
            if (!(mag_stash[t] = dalloc(mag_floats)))
                MG_FAIL(UMX_ERR_HIP, "out of device memory");
    }
    const int col_out = umx_plan::edge_colour(my_stage, pl.P), col_in = umx_plan::edge_colour((my_stage + pl.P - 1) % pl.P, pl.P);
    bool sent_valid[3] = {false, false, false}, mags_sent_valid = false;

    // ---- the segments of this rank, in order.  Everything is queued; nothing waits on the host.
    for (Local &lc : mine)
    {
        // the chunk of the (shifted) track: zeros outside [lead, lead + length)
        MG_HIP(hipMemsetAsync(lc.in, 0, sizeof(float) * 2 * (size_t)lc.n, st));
        const long long lo = std::max<long long>(lc.off, lead), hi = std::min<long long>((long long)lc.off + lc.n, (long long)lead + length);
        if (hi > lo)
            MG_HIP(hipMemcpyAsync(lc.in + 2 * (size_t)(lo - lc.off), audio_host + 2 * (size_t)(lo - lead), sizeof(float) * 2 * (size_t)(hi - lo),
                                  hipMemcpyHostToDevice, st));
        MG_UMX(umx_hip_segment_begin_device(ctx, lc.in, lc.n, eng_flags));
        for (int l = 0; l < 3; ++l)
        {
            if (lc.seg == 0) // umx.cpp:167-171: a track starts from zero state
                for (int t = 0; t < 4; ++t)
                    MG_HIP(hipMemsetAsync(state_of(t, l), 0, sizeof(float) * lstate, st));
            else if (loop)
            {
                // the state segment s - 1 left comes back out of the stash it was sent to (state_of is poisoned in between)
                if (m->hop_counter++ == m->debug_stall_hop) // testing: the stream stands still as if the peer's side of this hop never came
                    MG_HIP(hipLaunchHostFunc(st, debug_stall_fn, reinterpret_cast<void *>(static_cast<intptr_t>(2 * m->timeout_ms + 1000))));
                MG_NCCL(ncclGroupStart());
                for (int t = 0; t < 4; ++t)
                {
                    MG_NCCL(ncclSend(stash[l][t], lstate, ncclFloat, 0, m->ring[0], st));
                    MG_NCCL(ncclRecv(state_of(t, l), lstate, ncclFloat, 0, m->ring[0], st));
                }
                MG_NCCL(ncclGroupEnd());
                m->stats[0] += 8;
                m->stats[1] += 4;
            }
            else if (pl.P > 1)
            {
                // this rank's previous send of the layer's state read the buffer the receive is about to fill
                if (sent_valid[l])
                    MG_HIP(hipStreamWaitEvent(st, m->sent[l], 0));
                MG_NCCL(ncclGroupStart());
                for (int k = 0; k < n_mine; ++k)
                    MG_NCCL(ncclRecv(state_of(mine_t[k], l), lstate, ncclFloat, pl.prev_rank(rank, lc.seg), m->ring[col_in], st));
                MG_NCCL(ncclGroupEnd());
                m->stats[0] += n_mine;
                m->stats[1] += n_mine;
            } // P == 1: the state segment s - 1 left is already in place
            MG_UMX(umx_hip_segment_lstm_layer(ctx, l));
            if (lc.seg + 1 < nseg)
            {
                if (loop)
                {
                    MG_NCCL(ncclGroupStart());
                    for (int t = 0; t < 4; ++t)
                    {
                        MG_NCCL(ncclSend(state_of(t, l), lstate, ncclFloat, 0, m->ring[0], st));
                        MG_NCCL(ncclRecv(stash[l][t], lstate, ncclFloat, 0, m->ring[0], st));
                    }
                    MG_NCCL(ncclGroupEnd());
                    for (int t = 0; t < 4; ++t) // NaN bit patterns: a next segment that did not take the state from RCCL would show
                        MG_HIP(hipMemsetAsync(state_of(t, l), 0xFF, sizeof(float) * lstate, st));
                    m->stats[0] += 8;
                    m->stats[1] += 4;
                }
                else if (pl.P > 1)
                {
                    MG_HIP(hipEventRecord(m->layer_done[l], st));
                    MG_HIP(hipStreamWaitEvent(ss, m->layer_done[l], 0));
                    MG_NCCL(ncclGroupStart());
                    for (int k = 0; k < n_mine; ++k)
                        MG_NCCL(ncclSend(state_of(mine_t[k], l), lstate, ncclFloat, pl.next_rank(rank, lc.seg), m->ring[col_out], ss));
                    MG_NCCL(ncclGroupEnd());
                    MG_HIP(hipEventRecord(m->sent[l], ss));
                    sent_valid[l] = true;
                    m->stats[0] += n_mine;
                    m->stats[1] += n_mine;
                }
            }
        }
        // fc2, fc3 of this group's targets; the previous segment's magnitudes may still be on their way out
        if (mags_sent_valid)
            MG_HIP(hipStreamWaitEvent(st, m->mags_sent, 0));
        MG_UMX(umx_hip_segment_masks_device(ctx));
        for (int k = 0; k < n_mine; ++k) // a target the CALLER skipped (BASELINE config 1) is silence, as in umx_hip_infer_segment
            if (flags & UMX_FLAG_SKIP_TARGET(mine_t[k]))
                MG_HIP(hipMemsetAsync(mag[mine_t[k]], 0, sizeof(float) * mag_floats, st));
        const int wr = pl.wiener_rank(lc.seg);
        if (pl.G > 1 || loop)
        {
            MG_HIP(hipEventRecord(m->mark, st)); // everything of this rank up to here, incl. an earlier filter that read the buffers
            MG_HIP(hipStreamWaitEvent(ms, m->mark, 0));
            if (loop)
            {
                MG_NCCL(ncclGroupStart());
                for (int t = 0; t < 4; ++t)
                {
                    MG_NCCL(ncclSend(mag[t], mag_floats, ncclFloat, 0, m->mag_comm, ms));
                    MG_NCCL(ncclRecv(mag_stash[t], mag_floats, ncclFloat, 0, m->mag_comm, ms));
                }
                MG_NCCL(ncclGroupEnd());
                for (int t = 0; t < 4; ++t)
                    MG_HIP(hipMemsetAsync(mag[t], 0xFF, sizeof(float) * mag_floats, ms));
                MG_NCCL(ncclGroupStart());
                for (int t = 0; t < 4; ++t)
                {
                    MG_NCCL(ncclSend(mag_stash[t], mag_floats, ncclFloat, 0, m->mag_comm, ms));
                    MG_NCCL(ncclRecv(mag[t], mag_floats, ncclFloat, 0, m->mag_comm, ms));
                }
                MG_NCCL(ncclGroupEnd());
                m->stats[0] += 16;
                m->stats[2] += 8;
            }
            else if (wr != rank)
            {
                MG_NCCL(ncclGroupStart());
                for (int k = 0; k < n_mine; ++k)
                    MG_NCCL(ncclSend(mag[mine_t[k]], mag_floats, ncclFloat, wr, m->mag_comm, ms));
                MG_NCCL(ncclGroupEnd());
                MG_HIP(hipEventRecord(m->mags_sent, ms));
                mags_sent_valid = true;
                m->stats[0] += n_mine;
                m->stats[2] += n_mine;
            }
            else
            {
                MG_NCCL(ncclGroupStart());
                for (int t = 0; t < 4; ++t)
                    if (!pl.owns_target(rank, t))
                    {
                        MG_NCCL(ncclRecv(mag[t], mag_floats, ncclFloat, pl.owner_of_target(t, my_stage), m->mag_comm, ms));
                        ++m->stats[0];
                        ++m->stats[2];
                    }
                MG_NCCL(ncclGroupEnd());
            }
            if (wr == rank)
            {
                MG_HIP(hipEventRecord(m->mags_here, ms));
                MG_HIP(hipStreamWaitEvent(st, m->mags_here, 0));
            }
        }
        if (wr == rank)
        {
            MG_UMX(umx_hip_segment_finish_device(ctx, lc.stems)); // wiener_filter + istft (inference.cpp:192-207)
            MG_UMX(umx_hip_weight_stems_device(ctx, lc.stems, lc.n, st)); // umx.cpp:246
        }
        else
            MG_UMX(umx_hip_segment_discard(ctx));
        MG_HIP(hipEventRecord(lc.done, st));
    }

    // ---- gather: rank 0 adds the weighted stems in segment order (umx.cpp:234-260)
    float *track[4] = {}, *sumw = nullptr, *rbuf[2][4] = {};
    if (rank != 0)
    {
        for (Local &lc : mine)
            if (lc.stems[0])
            {
                MG_HIP(hipStreamWaitEvent(gs, lc.done, 0));
                MG_NCCL(ncclGroupStart());
                for (int t = 0; t < 4; ++t)
                    MG_NCCL(ncclSend(lc.stems[t], (size_t)2 * lc.n, ncclFloat, 0, m->gather_comm, gs));
                MG_NCCL(ncclGroupEnd());
                m->stats[0] += 4;
                m->stats[3] += 4;
            }
    }
    else
    {
        sumw = dalloc((size_t)L2);
        for (int t = 0; t < 4; ++t)
        {
            track[t] = dalloc((size_t)2 * L2);
            for (int b = 0; b < 2; ++b)
                rbuf[b][t] = (world > 1 || loop) ? dalloc((size_t)2 * N) : nullptr;
        }
        if (!sumw || !track[3] || ((world > 1 || loop) && !rbuf[1][3]))
            MG_FAIL(UMX_ERR_HIP, "out of device memory");
        for (int t = 0; t < 4; ++t)
            MG_HIP(hipMemsetAsync(track[t], 0, sizeof(float) * 2 * (size_t)L2, as));
        MG_HIP(hipMemsetAsync(sumw, 0, sizeof(float) * (size_t)L2, as));
        size_t mi = 0;
        bool added_valid[2] = {false, false};
        int nrecv = 0;
        for (int i = 0; i < nseg; ++i)
        {
            const int off = offsets[i], n = std::min(N, L2 - off), wr = pl.wiener_rank(i);
            Local *lc = nullptr;
            if (pl.runs_segment(0, i))
                lc = &mine[mi++];
            if (wr == 0 && !loop)
            {
                MG_HIP(hipStreamWaitEvent(as, lc->done, 0));
                MG_UMX(umx_hip_track_accumulate_device(ctx, track, sumw, lc->stems, off, n, as));
                continue;
            }
            // two receive buffers: the transfer of one segment's stems runs while the previous one's are being added
            const int b = nrecv++ & 1;
            float *const *rb = rbuf[b];
            if (added_valid[b])
                MG_HIP(hipStreamWaitEvent(gs, m->added[b], 0));
            if (loop)
                MG_HIP(hipStreamWaitEvent(gs, lc->done, 0));
            MG_NCCL(ncclGroupStart());
            for (int t = 0; t < 4; ++t)
            {
                if (loop)
                    MG_NCCL(ncclSend(lc->stems[t], (size_t)2 * n, ncclFloat, 0, m->gather_comm, gs));
                MG_NCCL(ncclRecv(rb[t], (size_t)2 * n, ncclFloat, loop ? 0 : wr, m->gather_comm, gs));
            }
            MG_NCCL(ncclGroupEnd());
            m->stats[0] += loop ? 8 : 4;
            m->stats[3] += 4;
            MG_HIP(hipEventRecord(m->recvd[b], gs));
            MG_HIP(hipStreamWaitEvent(as, m->recvd[b], 0));
            MG_UMX(umx_hip_track_accumulate_device(ctx, track, sumw, rb, off, n, as));
            MG_HIP(hipEventRecord(m->added[b], as));
            added_valid[b] = true;
        }
        MG_UMX(umx_hip_track_normalise_device(ctx, track, sumw, L2, as)); // umx.cpp:264-273
    }

    // ---- wait, then agree on how it went BEFORE rank 0 hands anything out: a persistent-LSTM timeout on one rank has sent
    // garbage state and stems on (every transfer still took place, so nobody hangs)
    if (int rc = wait_streams(m, {st, ss, ms, gs, as}, "the track's kernels and transfers", err))
        return rc;
    int local = umx_hip_sync(ctx); // (every stream is idle: this collects the engine's status, it does not wait)
    if (local != UMX_OK && local != UMX_ERR_TIMEOUT)
        MG_FAIL(local, std::string("umx_hip_sync: ") + umx_hip_last_error(ctx));
    int global = local;
    if (m->gather_comm)
    {
        MG_HIP(hipMemcpyAsync(m->status_dev, &local, sizeof(int), hipMemcpyHostToDevice, gs));
        MG_NCCL(ncclAllReduce(m->status_dev, m->status_dev, 1, ncclInt, ncclMax, m->gather_comm, gs));
        MG_HIP(hipMemcpyAsync(&global, m->status_dev, sizeof(int), hipMemcpyDeviceToHost, gs));
        if (int rc = wait_streams(m, {gs}, "the status all-reduce", err))
            return rc;
        ++m->stats[0];
    }
    *global_status = global;
    if (global != UMX_OK)
    {
        seterr(err, "a persistent LSTM launch timed out on " + std::string(local != UMX_OK ? "this rank" : "another rank"));
        return UMX_OK; // the caller decides (retry); nothing was written to out_host
    }
    if (rank == 0)
    {
        for (int t = 0; t < 4; ++t) // umx.cpp:136-147: drop the shift
            MG_HIP(hipMemcpyAsync(out_host[t], track[t] + 2 * (size_t)lead, sizeof(float) * 2 * (size_t)length, hipMemcpyDeviceToHost, as));
        if (int rc = wait_streams(m, {as}, "the download of the stems", err))
            return rc;
    }
    return UMX_OK;
}
} // namespace

extern "C" int umx_mgpu_separate_track(umx_mgpu *m, const float *audio_host, int length, int shift_offset,
                                       float *const out_host[4], unsigned flags, char *err)
{
    if (!m || !audio_host || length < 1 || shift_offset >= UMX_MAX_SHIFT || (m->rank == 0 && !out_host))
        MG_FAIL(UMX_ERR_ARG, "umx_mgpu_separate_track: bad argument");
    if (m->dead)
        MG_FAIL(UMX_ERR_HIP, "umx_mgpu_separate_track: the communicators were aborted after an earlier error");
    m->stats[4] = 0;
    m->hop_counter = 0;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        int global = UMX_OK;
        const int rc = separate_once(m, audio_host, length, shift_offset, out_host, flags & ~(attempt ? UMX_FLAG_DEBUG_LSTM_ABORT : 0u), &global, err);
        if (rc != UMX_OK)
        {
            // this rank cannot go on: its peers must not be left waiting on kernels of ours, and a device-wide wait could
            // itself hang on queued receives -- abort the communicators first
            m->abort_comms();
            (void)hipDeviceSynchronize();
            (void)umx_hip_segment_discard(m->ctx);
            return rc;
        }
        if (global == UMX_OK)
            return UMX_OK;
        ++m->stats[4]; // every rank saw the same status: all of them run the track again (the one that timed out on its per-step driver)
    }
    return UMX_ERR_TIMEOUT;
}
