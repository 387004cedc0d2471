// engine.hip -- C-ABI implementation (include/umx_hip.h) of the gfx950 UMX segment engine.
// Orchestrates one segment exactly like umx_inference (inference.cpp:12-207):
//   stft -> |.| / crop+stack -> 4 x [fc1 bn tanh -> 3-layer BiLSTM -> fc2 bn relu -> fc3 bn scale
//   relu -> mask*mix] -> Wiener EM -> 4 x istft
// The four targets run inside the same launches; consecutive segments alternate between two pipeline
// slots (streams) so that their LSTM layers overlap as an exact wavefront (see struct Slot).  A context created
// for several tracks (umx_hip_create_tracks) runs one segment of each track per call: every stage is queued for
// all track lanes, and the LSTM recurrence of all lanes is ONE launch (lstm_batch.h).  Also here: the
// whole-track drivers (split / shift inference with the track resident in HBM), the phased form of a segment
// for the multi-GPU state-carry mode, weight residency (u8/u16 as stored, or expanded) and the GEMM flavour.
#include "../../include/umx_hip.h"
#include <chrono>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "gemm_bf16x3.h"
#include "gemm_planes.h"
#include "gemm_planes_pp.h"
#include "lstm_kernels.h"
#include "lstm_batch.h"
#include "lstm_batch2.h"
#include "track_kernels.h"
#include "stft_kernels.h"
#include "wiener_kernels.h"
#include "wiener_istft.h"

using namespace umx;

// UMX_HIP_CHECK for the free functions of the C-ABI (the error goes to the context)
#define UMX_HIP_CHECK_CTX(ctx_, expr)                                                                                \
    do                                                                                                               \
    {                                                                                                                \
        hipError_t _e = (expr);                                                                                      \
        if (_e != hipSuccess)                                                                                        \
        {                                                                                                            \
            (ctx_)->set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                                    \
            return UMX_ERR_HIP;                                                                                      \
        }                                                                                                            \
    } while (0)

namespace
{
std::string g_create_error = "";

// A GEMM weight kept in HBM as stored in the ggml file (BASELINE config 5): u8 / u16 + (scale, offset);
// two parameter pairs because W_ih of a layer is two file tensors (forward rows, then reverse rows).
struct QMat
{
    void *q = nullptr;
    int type = 0; // GemmBType
    float s[2] = {1.f, 1.f}, o[2] = {0.f, 0.f};
};

// A GEMM weight as fp16 planes [nbp][N][K] (gemm_planes.h): u8 -> 1 exact plane of q - 128, u16 -> 2 planes whose sum is
// exactly q - 32896: fp16(q - 32896) and the remainder (an integer of at most 16), fp32 -> 2 split terms of w / s with s a
// power of two; (scale, offset + c scale) per file tensor (W_ih: two).
struct PMat
{
    unsigned short *p = nullptr;
    int nbp = 2;
    float s[2] = {1.f, 1.f}, o2[2] = {0.f, 0.f};
};

struct TargetBufs // weights of one target (shared by both pipeline slots)
{
    PMat fc1_p, ih_p[3], fc2_p, fc3_p; // gemm_planes.h (the default GEMM flavour)
    QMat fc1_q, ih_q[3], fc2_q, fc3_q; // used instead of the fp32 matrix when .q != nullptr
    unsigned short *fc1_bx = nullptr, *ih_bx[3] = {}, *fc2_bx = nullptr, *fc3_bx = nullptr; // bf16 planes [3][N][K] (gemm_bf16x3.h)
    float *fc1_w = nullptr, *in_scale = nullptr, *in_mean = nullptr, *bn1[4] = {};
    float *ih_w[3] = {}, *ih_b[3] = {};
    float *fc2_w = nullptr, *bn2[4] = {};
    float *fc3_w = nullptr, *bn3[4] = {}, *out_scale = nullptr, *out_mean = nullptr;
};

struct TargetAct // activations of one target in one pipeline slot
{
    float *cat = nullptr, *la = nullptr, *lb = nullptr, *P = nullptr, *a2 = nullptr,
          *mag = nullptr; // mag: fc3's MASK [2][T][MAGP]; x |X| = the target magnitude, formed by the consumers (gemm_common.h)
    // gemm_planes.h: every GEMM's A operand split once into two fp16 planes of the scaled row, its row sums and (for the
    // unbounded tensors) its per-row inverse scales
    unsigned short *xs_p = nullptr, *cat_p = nullptr, *la_p = nullptr, *lb_p = nullptr, *a2_p = nullptr;
    float *rs_xs = nullptr, *rs_catL = nullptr, *rs_catR = nullptr, *rs_la = nullptr, *rs_lb = nullptr, *rs_a2 = nullptr;
    float *rsc_xs = nullptr, *rsc_a2 = nullptr;
};

enum
{
    ST_STFT = 0,
    ST_FC1,
    ST_IH0,
    ST_LSTM0,
    ST_IH1,
    ST_LSTM1,
    ST_IH2,
    ST_LSTM2,
    ST_FC2,
    ST_FC3,
    ST_WIENER,
    ST_ISTFT,
    ST_OLA,
    ST_COUNT
};
enum { SP_XS = 0, SP_CATL, SP_LA, SP_LB, SP_CATR, SP_A2 }; // which A operand launch_split prepares
const char *kStageNames[ST_COUNT] = {"stft",  "fc1", "lstm_ih0", "lstm_rec0", "lstm_ih1", "lstm_rec1", "lstm_ih2",
                                     "lstm_rec2", "fc2", "fc3_mask", "wiener",  "istft",    "ola"};

// roctx ranges per stage (tracing): every stage marker below also opens a named range on the calling thread
// ("umx:<stage>": the HOST span in which the stage's kernels are queued; rocprofv3 --marker-trace shows them beside the
// kernel trace).  The marker library is looked up at run time (rocprofiler-sdk's, then roctracer's): no link dependency,
// and without a profiler attached the calls are a null-pointer test.
namespace
{
struct RoctxApi
{
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"})
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))
            {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop)
                    return;
                push = nullptr;
                pop = nullptr;
            }
    }
};
thread_local bool t_stage_range_open = false;
// closes the calling thread's open stage range and opens `stage` (< 0: only closes)
inline void stage_range(int stage)
{
    static const RoctxApi api;
    if (!api.push)
        return;
    if (t_stage_range_open)
        api.pop();
    t_stage_range_open = stage >= 0 && stage < ST_COUNT;
    if (t_stage_range_open)
    {
        char name[40];
        snprintf(name, sizeof name, "umx:%s", kStageNames[stage]);
        api.push(name);
    }
}
} // namespace

// One track lane of a pipeline slot = the activations of one in-flight segment of one track.
struct Lane
{
    TargetAct ta[4];
    float2 *spec = nullptr, *y = nullptr, *frames = nullptr;
    float *x = nullptr, *wpart = nullptr, *Rc = nullptr;
    unsigned *maxabs = nullptr;
};

// One pipeline slot = everything one in-flight segment (of every track lane) needs.  Two slots on two streams let
// segment s+1 run its STFT/GEMMs and LSTM layer l while segment s is in layer l+1 (the exact wavefront of
// R_l(s+1) waits only for R_l(s) through an event); the single-track LSTM kernels are latency-bound
// and use half of each CU's wave slots, so two of them co-reside and fill each other's hand-off gaps.
struct Slot
{
    hipStream_t stream = nullptr;
    Lane lane[LSTMB_MAX_TRACKS];
    float *hbuf = nullptr;
    bool lstm_wrote_planes[3] = {false, false, false}; // this call's recurrence of layer l wrote the next GEMM's A planes (run_lstm_layer_batched)
    unsigned *status = nullptr, *lsync = nullptr;
    unsigned long long *lprof = nullptr;
    hipEvent_t ev[ST_COUNT + 1] = {};
    hipEvent_t evk[ST_COUNT] = {}; // GEMM stages of plane contexts: recorded between the stage's split kernel and its GEMM kernel
    bool evk_set[ST_COUNT] = {};
    hipEvent_t rec_done[3] = {}; // LSTM layer l of this slot's segment has finished (state updated)
    // host-pointer entry points (see umx_hip_infer_batch_async): the download of call k's stems is queued on the OTHER slot's
    // stream, behind call k + 1's kernels
    hipEvent_t k_done = nullptr;   // this slot's stems are complete (recorded on its own stream)
    hipEvent_t out_free = nullptr; // their download has finished (recorded on the other slot's stream): overlap-add may overwrite them
    bool out_free_valid = false;
    bool have_times = false, last_persistent = false, used = false;
};
} // namespace

// Admission of persistent LSTM grids, process-wide and per device.  A persistent launch needs ALL its workgroups
// co-resident (they exchange granules), so the grids that can be resident TOGETHER must fit the device together.
// Grids queued on one stream run one after the other, so a stream contributes at most its largest queued grid; what has
// to fit is the sum over streams.  A launch that would not fit is made to wait -- on the device, through a stream-wait
// on another stream's newest admitted grid (which, streams being in order, is a wait for all of that stream's grids),
// never by blocking the host -- for the streams whose queued grids are oldest, until it fits.  The streams it waits for
// hold only grids admitted earlier, i.e. work that does not depend on the new launch: no cycle.
// (Rounds 1-2 summed over all queued grids instead of over streams: with the host running ahead of the device, a launch then
// waited for every earlier grid but the most recent one, and the two-slot wavefront of a single-track context overlapped
// only L2(s) with L0(s+1) -- one pair per segment instead of every layer.  Found in round 3.)
// Units are half CUs: a single-track workgroup (two fit a CU) counts 1, a batched one (one per CU) counts 2.
namespace
{
struct LstmGate
{
    std::mutex m;
    struct Grid
    {
        hipEvent_t done;
        int units;
        hipStream_t stream;
        unsigned long long seq;
    };
    std::vector<Grid> inflight; // in admission order
    std::vector<hipEvent_t> pool;
    unsigned long long seq = 0;
    int reserved = 0; // half CUs kept free for kernels that are not this engine's (RCCL send / recv: umx_hip_gate_reserve)
    std::vector<int> reservations; // outstanding requests in CUs: `reserved` follows the largest
};
LstmGate g_gate[16];

// admit + launch + record under the gate's lock (an event that has not been recorded yet would read as complete)
template <class Launch> hipError_t lstm_gate_launch(int device, hipStream_t st, int units, int capacity_units, Launch launch)
{
    LstmGate &g = g_gate[device & 15];
    std::lock_guard<std::mutex> lock(g.m);
    capacity_units -= g.reserved;
    for (size_t i = 0; i < g.inflight.size();)
        if (hipEventQuery(g.inflight[i].done) == hipSuccess)
        {
            g.pool.push_back(g.inflight[i].done);
            g.inflight.erase(g.inflight.begin() + i);
        }
        else
            ++i;
    (void)hipGetLastError(); // hipEventQuery reports "not ready" as an error
    // per other stream: its largest queued grid, the admission number of its oldest one, and its newest grid's event
    struct PerStream
    {
        hipStream_t stream;
        int units;
        unsigned long long oldest;
        hipEvent_t newest;
    };
    std::vector<PerStream> others;
    for (const LstmGate::Grid &gr : g.inflight)
    {
        if (gr.stream == st)
            continue;
        PerStream *ps = nullptr;
        for (PerStream &o : others)
            if (o.stream == gr.stream)
                ps = &o;
        if (!ps)
        {
            others.push_back({gr.stream, 0, gr.seq, gr.done});
            ps = &others.back();
        }
        ps->units = std::max(ps->units, gr.units);
        ps->newest = gr.done; // admission order: the last one seen is the newest
    }
    int used = 0;
    for (const PerStream &o : others)
        used += o.units;
    while (used + units > capacity_units && !others.empty())
    {
        size_t k = 0;
        for (size_t i = 1; i < others.size(); ++i)
            if (others[i].oldest < others[k].oldest)
                k = i;
        (void)hipStreamWaitEvent(st, others[k].newest, 0); // every grid of that stream has left the device when we start
        used -= others[k].units;
        others.erase(others.begin() + k);
    }
    hipEvent_t ev = nullptr;
    if (!g.pool.empty())
    {
        ev = g.pool.back();
        g.pool.pop_back();
    }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
        ev = nullptr;
    const hipError_t e = launch();
    if (ev && e == hipSuccess && hipEventRecord(ev, st) == hipSuccess)
        g.inflight.push_back({ev, units, st, g.seq++});
    else if (ev)
        g.pool.push_back(ev);
    return e;
}
} // namespace

// kernel instantiation pickers
static const void *lstm_persistent_fn(int kpw, bool precise)
{
    return precise ? (kpw == 8    ? reinterpret_cast<const void *>(lstm_persistent_kernel<8, true>)
                      : kpw == 16 ? reinterpret_cast<const void *>(lstm_persistent_kernel<16, true>)
                      : kpw == 32 ? reinterpret_cast<const void *>(lstm_persistent_kernel<32, true>)
                                  : reinterpret_cast<const void *>(lstm_persistent_kernel<64, true>))
                   : (kpw == 8    ? reinterpret_cast<const void *>(lstm_persistent_kernel<8, false>)
                      : kpw == 16 ? reinterpret_cast<const void *>(lstm_persistent_kernel<16, false>)
                      : kpw == 32 ? reinterpret_cast<const void *>(lstm_persistent_kernel<32, false>)
                                  : reinterpret_cast<const void *>(lstm_persistent_kernel<64, false>));
}
template <int HL> static const void *lstm_batch_fn_hl(bool wq, bool precise)
{
    return wq ? (precise ? reinterpret_cast<const void *>(lstm_batch_kernel<HL, true, true>)
                         : reinterpret_cast<const void *>(lstm_batch_kernel<HL, true, false>))
              : (precise ? reinterpret_cast<const void *>(lstm_batch_kernel<HL, false, true>)
                         : reinterpret_cast<const void *>(lstm_batch_kernel<HL, false, false>));
}
template <int G> static const void *lstm_batch2_fn_g(int Hl, bool precise)
{
    switch (Hl)
    {
    case 64: return precise ? reinterpret_cast<const void *>(lstm_batch2_kernel<64, G, true>) : reinterpret_cast<const void *>(lstm_batch2_kernel<64, G, false>);
    case 128: return precise ? reinterpret_cast<const void *>(lstm_batch2_kernel<128, G, true>) : reinterpret_cast<const void *>(lstm_batch2_kernel<128, G, false>);
    case 256: return precise ? reinterpret_cast<const void *>(lstm_batch2_kernel<256, G, true>) : reinterpret_cast<const void *>(lstm_batch2_kernel<256, G, false>);
    case 512: return precise ? reinterpret_cast<const void *>(lstm_batch2_kernel<512, G, true>) : reinterpret_cast<const void *>(lstm_batch2_kernel<512, G, false>);
    default: return nullptr;
    }
}
// two or three groups of 16 lanes (lstm_batch2.h): u8-resident W_hh only
static const void *lstm_batch2_fn(int Hl, int groups, bool precise) { return groups == 3 ? lstm_batch2_fn_g<3>(Hl, precise) : lstm_batch2_fn_g<2>(Hl, precise); }
#ifndef UMX_FUSE_LSTM_PLANES
#define UMX_FUSE_LSTM_PLANES 1
#endif
static int lstmb2_bulk(int groups) { return groups == 3 ? 2 : 4; } // ring rows per fetch: what fits the LDS beside the partial sums
// two groups of 16 lanes side by side on the chip, chains of 16 workgroups with two slices each (lstm_batchs_kernel): hidden 512 / 1024,
// u8-resident W_hh
static const void *lstm_batchs_fn(int Hl, int groups, bool precise)
{
    if (groups != 2)
        return nullptr;
    switch (Hl)
    {
    case 256: return precise ? reinterpret_cast<const void *>(lstm_batchs_kernel<256, true, 2>) : reinterpret_cast<const void *>(lstm_batchs_kernel<256, false, 2>);
    case 512: return precise ? reinterpret_cast<const void *>(lstm_batchs_kernel<512, true, 2>) : reinterpret_cast<const void *>(lstm_batchs_kernel<512, false, 2>);
    default: return nullptr;
    }
}
constexpr int kBatchsBulk = 1, kBatchsSpan = 2; // ring rows per fetch (LDS: 128 KB of partial sums + 2 rows x 16 lanes x 528 B), slices per workgroup
static const void *lstm_batch_fn(int Hl, bool wq, bool precise)
{
    switch (Hl)
    {
    case 64: return lstm_batch_fn_hl<64>(wq, precise);
    case 128: return lstm_batch_fn_hl<128>(wq, precise);
    case 256: return lstm_batch_fn_hl<256>(wq, precise);
    case 512: return lstm_batch_fn_hl<512>(wq, precise);
    default: return nullptr;
    }
}

struct umx_hip_ctx
{
    int device = 0, H = 0, Hl = 0, S = 0, N = 0, T = 0, Tp = 0, nbatch = 0;
    static constexpr int Mpad = 256; // slack rows behind the last lane: a launch's M is rounded up to the 256-row tile
    std::string err;
    std::vector<void *> allocs;
    TargetBufs tb[4];
    float *whh[3] = {}, *bhh[3] = {};
    float *window = nullptr, *nw = nullptr;
    float2 *tw1 = nullptr, *tw2 = nullptr;
    float *tap_tmp = nullptr;                    // umx_hip_read_tap: scratch of the computed taps
    float *audio_in = nullptr, *out_dev[4] = {}; // device staging of the phased (multi-GPU carry) entry points
    static constexpr int kMaxSlots = 3;
    float *stage_in[kMaxSlots] = {}, *stage_out[kMaxSlots][4 * LSTMB_MAX_TRACKS] = {}; // per pipeline slot: device staging of the host-pointer
    int ensure_staging();                                              // entry points, [lane] / [lane][4]; allocated on first use
    hipEvent_t order_ev = nullptr;
    struct DeferredDownload // the stems of a host-pointer call, in its slot's staging buffers
    {
        bool valid = false;
        int si = 0, nb = 0, n[LSTMB_MAX_TRACKS] = {};
        float *host[4 * LSTMB_MAX_TRACKS] = {};
    };
    int queue_download(const DeferredDownload &d, hipStream_t on); // D2H copies of d onto stream `on`, then out_free of its slot
    hipStream_t copy_stream = nullptr; // downloads of the host-pointer calls (created with the staging buffers)
    float *state = nullptr, *state_alt = nullptr; // state_alt: second copy for the per-step driver of the batched recurrence
    Slot slot[kMaxSlots];
    int nslots = 2; // pipeline slots: 3 for single-track contexts (see init), 2 for track-batched ones
    int next_slot() const { return (int)(nseg % nslots); }
    void clear_used()
    {
        for (int si = 0; si < kMaxSlots; ++si)
            slot[si].used = false;
    }
    int B = 1;                // track lanes (umx_hip_create_tracks); every lane has its own streaming LSTM state
    bool lstm_batched = false; // LSTM recurrence on the matrix cores for all lanes at once (lstm_batch.h); fixed at
                               // create so that a track's bits never depend on how many lanes a call uses
    unsigned tag_epoch = 0;   // persistent LSTM launches so far (granule tags are unique per launch)
    unsigned next_tag_base()
    {
        // 4096 tags per launch (T + 1 <= 4096 is checked at create); wraps after ~1M launches, where one
        // stale line from exactly 2^20 launches ago would have to survive in an L2 -- every buffer is
        // zeroed at that point anyway (see run_lstm_layer)
        tag_epoch = (tag_epoch + 1) & 0xFFFFF;
        return tag_epoch << 12;
    }
    int cur = 0;              // slot of the most recently queued segment
    long long nseg = 0;       // segments queued since creation
    size_t lsync_words = 0;
    bool persistent_ok = true;
    int n_cus = 256;
    // what umx_hip_sync needs to redo the calls queued since the last sync after a persistent-kernel timeout
    struct PendingCall
    {
        int nb;
        const float *audio[LSTMB_MAX_TRACKS];
        int n[LSTMB_MAX_TRACKS];
        float *out[4 * LSTMB_MAX_TRACKS];
        float *host_out[4 * LSTMB_MAX_TRACKS]; // host-pointer entry points: where the stems are copied afterwards
        const float *host_audio[LSTMB_MAX_TRACKS]; // ... and where the audio came from: the two staging buffers have been
                                                   // reused by later calls, so a replay uploads it again
        unsigned flags;
    };
    static constexpr int kBackupCalls = 8;
    std::vector<PendingCall> pending; // at most kBackupCalls entries
    bool pending_lost = false;        // calls were queued that cannot be replayed: more than kBackupCalls since the last sync,
                                      // or the caller was released from the buffer contract (umx_hip_order_before)
    float *backup = nullptr; // [kBackupCalls][3 layers][B * state_floats]: the stream state right before each layer launch
    bool no_recovery = false, recovering = false;
    int recover();
    int lstm_poll_delay = 0;         // 0 = the kernel's default (LSTM_POLL_DELAY)
    int lstm_threads = LSTM_THREADS; // 512 (two workgroups per CU fit) or 576 (dedicated gate wave)
    int lstm_capacity = 0;           // workgroups of the persistent LSTM kernel that can be co-resident
    unsigned last_flags = 0;
    hipStream_t stream = nullptr; // = slot[0].stream (H2D/D2H of the host-pointer entry point)

    void set_error(const std::string &s) { err = s; }

    template <class T_> int dalloc(T_ **p, size_t count, bool zero = true)
    {
        void *q = nullptr;
        UMX_HIP_CHECK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T_)));
        allocs.push_back(q);
        if (zero)
            UMX_HIP_CHECK(hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T_)));
        *p = reinterpret_cast<T_ *>(q);
        return UMX_OK;
    }
    template <class T_> int upload(T_ **p, const std::vector<T_> &h)
    {
        int rc = dalloc(p, h.size(), false);
        if (rc)
            return rc;
        UMX_HIP_CHECK(hipMemcpy(*p, h.data(), h.size() * sizeof(T_), hipMemcpyHostToDevice));
        return UMX_OK;
    }
    int init(int device_, int hidden, int segment_samples, const umx_tensor_view *tensors, int n_tensors,
             unsigned create_flags, int n_tracks);
    size_t weight_bytes = 0;      // HBM held by model tensors (the config-5 figure of merit)
    bool gemm_bf16x3 = false;     // dense stack on the bf16 matrix cores, three-term split (gemm_bf16x3.h)
    bool wiener_fused = true;     // wiener_istft.h: gains + filter + inverse STFT frame in one kernel
    bool gemm_planes = false;     // ... with both operands pre-split / re-encoded as bf16 planes and LDS-DMA staging (gemm_planes.h)
    void launch_split(Lane &ln, int nl, hipStream_t st, int which, const int *active, int nact);
    void launch_gemm_planes(Lane &ln, int nl, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg);
    // one GEMM stage for the track lanes with audio: the plane GEMMs take every run of consecutive lanes in one launch
    void launch_gemm_lanes(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, int mode, int layer, const int *active,
                           int nact, bool dbg);
    bool u8_dequant = false;      // UMX_CREATE_U8_DEQUANT: u8 weights dequantised per element (model.cpp:610-616) before they
                                  // are multiplied, instead of exact bf16 integers with the affine map applied to the sum
    unsigned char *whh_q[3] = {}; // u8-resident W_hh (create flag), same layout as whh[]
    float whh_s[3][8] = {}, whh_o[3][8] = {};
    int infer_device(const float *audio_dev, int n, float *const out[4], unsigned flags);
    // one segment of each of `nb` track lanes (lane i = track i of this context; audio[i] == nullptr: lane idle)
    int infer_batch(int nb, const float *const *audio_dev, const int *n, float *const *out /* [nb][4] */, unsigned flags);
    int lstm_batch_capacity = 0; // co-resident workgroups of the batched LSTM kernel
    bool lstm_batchs_ok = false; // lstm_batchs_kernel (two groups of 16 lanes side by side, 16 workgroups per chain) fits the chip
    bool lstm_rowsums = false;       // the batched recurrence hands the consuming plane GEMM the row sums of its output (lstm_batch.h, LstmBArgs::rs_dir)
    bool lstm_writes_planes = false; // ... and writes that GEMM's A planes itself (contexts of up to 32 lanes: lstm_batch_kernel / lstm_batchs_kernel)
    size_t state_floats() const { return (size_t)4 * 12 * Hl; }
    // phased form of one segment (exact multi-GPU carry): front | layer 0 | layer 1 | layer 2 | back
    // whole track on the device (split_inference / shift_inference, umx.cpp:99-295)
    int track(const float *audio_host, int length, int shift_offset, float *const out_host[4], unsigned flags,
              void (*progress)(float, void *), void *progress_user);
    int tracks(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host, unsigned flags,
               void (*progress)(float, void *), void *progress_user);
    int tracks_once(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host,
                    unsigned flags, void (*progress)(float, void *), void *progress_user);
    struct TrackBufs // whole-track driver, per track lane: the (shifted) track, 4 stem accumulators, weight sum, 2 x 4 segment stems
    {
        float *in = nullptr, *out[4] = {}, *sumw = nullptr, *seg[kMaxSlots][4] = {};
        size_t cap = 0;
    };
    std::vector<TrackBufs> trk;
    hipEvent_t trk_acc_ev[kMaxSlots] = {};
    int phase_begin(const float *audio_host, int n, unsigned flags);
    int phase_begin_device(const float *audio_dev, int n, unsigned flags);
    int phase_end_device(float *const out_dev_[4]);
    const float *ph_audio = nullptr;
    int phase_layer(int layer);
    int phase_end(float *const out_host[4]);
    int ph_next = -1; // -1: no phased segment open; 0..2: next LSTM layer; 3: back stage pending
    int ph_n = 0;
    unsigned ph_flags = 0;
    int run_lstm_layer(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask);
    int run_lstm_layer_batched(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask);
    int sync_all();
    void launch_gemm(Lane &ln, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg);
    int stage_front(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, const int *n, const int *active, int nact);
    // elements between the streaming buffers of consecutive track lanes of a slot (the kernels take lane 0's pointers)
    WienerStrides lane_strides() const
    {
        WienerStrides ls;
        const size_t nchunk = (size_t)(T + WIENER_CHUNK - 1) / WIENER_CHUNK;
        ls.spec = (size_t)2 * T * NBINS;
        ls.mag = (size_t)2 * T * MAGP;
        ls.part = std::max((size_t)4 * nbatch * NBINS * 9, nchunk * 4 * 5 * NBINS);
        ls.rc = (size_t)4 * NBINS * 4;
        ls.frames = (size_t)4 * T * NFFT;
        ls.y = (size_t)4 * 2 * T * NBINS;
        return ls;
    }
    static LaneSet lane_set(int nb, const float *const *audio_dev) // the active lanes of a call
    {
        LaneSet l;
        l.count = 0;
        for (int ln = 0; ln < nb; ++ln)
            if (audio_dev[ln])
                l.id[l.count++] = (unsigned char)ln;
        return l;
    }
    int stage_back(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n,
                   unsigned flags, const int *active, int nact);
    // its two halves: fc2 + fc3 -> the target magnitudes of the active targets | Wiener (or mixture phase), inverse STFT,
    // overlap-add from the magnitudes of ALL four targets (zero_skipped: a skipped target counts as silence; false when
    // its magnitudes were put there by someone else -- the target-sharded multi-GPU driver)
    int stage_masks(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, unsigned flags, const int *active, int nact);
    int stage_finish(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n, unsigned flags,
                     bool zero_skipped);
    int phase_masks();
    int phase_finish_device(float *const out_dev_[4]);
    static void active_list(unsigned flags, int *active, int &nact)
    {
        nact = 0;
        for (int tg = 0; tg < 4; ++tg)
            if (!(flags & UMX_FLAG_SKIP_TARGET(tg)))
                active[nact++] = tg;
    }
};

// ---------------------------------------------------------------- weights
namespace
{
// model.cpp:578-665: q*scale+offset in fp32; F32 passes through
bool dequant(const umx_tensor_view &tv, size_t expect, std::vector<float> &out)
{
    size_t nel = 1;
    for (int i = 0; i < tv.n_dims; ++i)
        nel *= (size_t)tv.ne[i];
    if (nel != expect)
        return false;
    out.resize(nel);
    if (tv.dtype == UMX_DTYPE_F32)
        memcpy(out.data(), tv.data, nel * sizeof(float));
    else if (tv.dtype == UMX_DTYPE_U8)
    {
        const uint8_t *q = static_cast<const uint8_t *>(tv.data);
        for (size_t i = 0; i < nel; ++i)
            out[i] = (float)q[i] * tv.scale + tv.offset;
    }
    else if (tv.dtype == UMX_DTYPE_U16)
    {
        const uint16_t *q = static_cast<const uint16_t *>(tv.data);
        for (size_t i = 0; i < nel; ++i)
            out[i] = (float)q[i] * tv.scale + tv.offset;
    }
    else
        return false;
    return true;
}
} // namespace

int umx_hip_ctx::init(int device_, int hidden, int segment_samples, const umx_tensor_view *tensors, int n_tensors,
                      unsigned create_flags, int n_tracks)
{
    if (n_tracks < 1 || n_tracks > LSTMB_MAX_TRACKS)
    {
        set_error("n_tracks must be in [1, 48]");
        return UMX_ERR_ARG;
    }
    B = n_tracks;
    lstm_batched = B > 1 || (create_flags & UMX_CREATE_LSTM_BATCHED);
    u8_dequant = create_flags & UMX_CREATE_U8_DEQUANT;
    if (hidden <= 0 || hidden % 128 != 0 || hidden > 2048)
    {
        set_error("hidden_size must be a positive multiple of 128 (<= 2048)");
        return UMX_ERR_ARG;
    }
    if (segment_samples < NFFT || segment_samples / HOP + 2 > 4096)
    {
        set_error("segment_samples must be in [4096, 4,190,000] (at most 4094 STFT frames per segment)");
        return UMX_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        set_error("no HIP device available (this engine has no CPU fallback)");
        return UMX_ERR_NODEVICE;
    }
    if (device_ < 0 || device_ >= ndev)
    {
        set_error("device index out of range");
        return UMX_ERR_ARG;
    }
    device = device_;
    UMX_HIP_CHECK(hipSetDevice(device));
    H = hidden;
    Hl = H / 2;
    S = Hl / LSTM_UNITS_PER_WG;
    N = segment_samples;
    T = N / HOP + 1; // dsp.hpp:48
    // rows per track lane of the lane-contiguous activation buffers.  Plane GEMMs run over all lanes at once and their
    // tiles may straddle lanes, so lanes follow each other without padding (T rows; 8 % fewer tile rows than lanes padded
    // to the 256-row tile) and only the END of a launch is padded to a tile (Mpad rows of slack in every buffer);
    // the per-lane GEMM flavours launch M = Tp per lane and need whole 128-row tiles.
    Tp = T; // provisional: fixed below once the GEMM flavour is known
    nbatch = (T + WIENER_BATCH - 1) / WIENER_BATCH;

    // ---- index the tensor views by (target, name)
    std::map<std::string, const umx_tensor_view *> idx[4];
    for (int i = 0; i < n_tensors; ++i)
    {
        if (tensors[i].target < 0 || tensors[i].target > 3 || !tensors[i].name || !tensors[i].data)
        {
            set_error("tensor view with bad target / name / data");
            return UMX_ERR_MODEL;
        }
        idx[tensors[i].target][tensors[i].name] = &tensors[i];
    }
    auto get = [&](int tg, const std::string &name, size_t expect, std::vector<float> &out) -> bool {
        auto it = idx[tg].find(name);
        if (it == idx[tg].end())
        {
            set_error("missing tensor '" + name + "' for target " + std::to_string(tg));
            return false;
        }
        if (!dequant(*it->second, expect, out))
        {
            set_error("tensor '" + name + "' has wrong size in model (target " + std::to_string(tg) + ")");
            return false;
        }
        return true;
    };

    const bool keepq = !(create_flags & UMX_CREATE_DEQUANTISE_AT_LOAD); // u8/u16 views stay as they are (config 5)
    auto view = [&](int tg, const std::string &name) -> const umx_tensor_view * {
        auto it = idx[tg].find(name);
        return it == idx[tg].end() ? nullptr : it->second;
    };
    auto nelems = [](const umx_tensor_view *tv) {
        size_t n = 1;
        for (int i = 0; i < tv->n_dims; ++i)
            n *= (size_t)tv->ne[i];
        return n;
    };
    // Upload `rows` x `cols` of a u8/u16 tensor as stored, into a (rows_pad x cols_pad) device matrix; padding
    // is q = 0 (any finite weight is fine there: padded K columns meet zero activations, padded N rows are
    // never stored).  rowmap (optional) = source row of each destination row.
    auto upload_q = [&](void **dst, const umx_tensor_view *tv, int rows, int cols, int rows_pad, int cols_pad,
                        const std::vector<int> *rowmap, size_t dst_row0, size_t total_rows) -> int {
        const size_t esz = tv->dtype == UMX_DTYPE_U8 ? 1 : 2;
        if (!*dst)
        {
            void *q = nullptr;
            UMX_HIP_CHECK(hipMalloc(&q, total_rows * cols_pad * esz));
            UMX_HIP_CHECK(hipMemset(q, 0, total_rows * cols_pad * esz));
            allocs.push_back(q);
            *dst = q;
            weight_bytes += total_rows * cols_pad * esz;
        }
        std::vector<unsigned char> host((size_t)rows_pad * cols_pad * esz, 0);
        const unsigned char *src = static_cast<const unsigned char *>(tv->data);
        for (int r = 0; r < rows; ++r)
        {
            const int sr = rowmap ? (*rowmap)[r] : r;
            memcpy(&host[(size_t)r * cols_pad * esz], src + (size_t)sr * cols * esz, (size_t)cols * esz);
        }
        UMX_HIP_CHECK(hipMemcpy(static_cast<unsigned char *>(*dst) + dst_row0 * cols_pad * esz, host.data(), host.size(),
                                hipMemcpyHostToDevice));
        return UMX_OK;
    };
    auto is_q = [&](const umx_tensor_view *tv, int dtype, size_t expect) {
        return keepq && tv && tv->dtype == dtype && nelems(tv) == expect;
    };
    // A GEMM launch covers all four targets with ONE kernel instantiation (its B-operand type is a template
    // parameter), so a matrix stays quantised only if it is stored that way for EVERY target; otherwise it is expanded
    // for all of them.
    auto all_q = [&](const std::string &name, int dtype, size_t expect) {
        for (int tg = 0; tg < 4; ++tg)
            if (!is_q(view(tg, name), dtype, expect))
                return false;
        return true;
    };

    if (create_flags & UMX_CREATE_GEMM_F32)
    {
        set_error("UMX_CREATE_GEMM_F32: the fp32-MFMA GEMM flavour was removed in round 3 (slower and, against float64, less accurate "
                  "than the split-operand kernels: profiles/r02_accuracy_vs_float64.txt)");
        return UMX_ERR_ARG;
    }
    gemm_bf16x3 = true; // 16-bit matrix cores with split operands: gemm_planes.h or gemm_bf16x3.h
    // Track-batched contexts fuse the Wiener filter with the inverse STFT (wiener_istft.h: one 1024-thread, 136 KB-LDS
    // workgroup per frame); the single-track context keeps the small kernels, which run beside the other slot's LSTM
    // grids (measured: fused 7.85 ms per segment in the pipeline, unfused 7.41).  UMX_WIENER = fused | stats4 | unfused.
    if (const char *e = getenv("UMX_LSTM_POLL_DELAY")) // tuning: x64 cycles a dot wave of the one-track recurrence sleeps before its first poll
        lstm_poll_delay = atoi(e);
    wiener_fused = lstm_batched;
    if (const char *e = getenv("UMX_WIENER")) // fused | stats4 (= statistics kernel + separate filter and inverse-STFT kernels)
        wiener_fused = std::string(e) == "fused";
    // gemm_planes.h for track-batched contexts (large tiles over all lanes); gemm_bf16x3.h for the single-track,
    // latency-optimised context, whose pipeline overlaps small GEMM blocks with two co-resident LSTM grids (the register
    // and LDS budget of DESIGN 4.2 was tuned for exactly that kernel).  Either can be forced.
    gemm_planes = gemm_bf16x3 && ((create_flags & UMX_CREATE_GEMM_PLANES) || (lstm_batched && !(create_flags & UMX_CREATE_GEMM_STAGED)));
    const bool bx = gemm_bf16x3;
    Tp = gemm_planes ? std::max(T, 256) : round_up(T, 128); // (a tile must not hold rows of more than two lanes)
    // A GEMM weight as fp16 planes (PMat): (rows x cols) of `tv` (u8 / u16 as stored: exact integers) or of `f32` (two
    // split terms of w * 2^e, 2^e bringing the tensor's largest |w| into [2^14, 2^15); returns 2^-e), source row
    // rowmap[r] -> destination row dst_row0 + r of a [nbp][total_rows][cols_pad] matrix built in `host`
    auto fill_planes = [&](std::vector<unsigned short> &host, int nbp, size_t total_rows, int cols_pad, const umx_tensor_view *tv,
                           const float *f32, int rows, int cols, const std::vector<int> *rowmap, size_t dst_row0) -> float {
        const size_t plane = total_rows * (size_t)cols_pad;
        if (host.empty())
            host.assign((size_t)nbp * plane, 0);
        float scale = 1.f, unscale = 1.f;
        if (f32)
        {
            float mx = 0.f;
            for (size_t i = 0; i < (size_t)rows * cols; ++i)
                if (std::isfinite(f32[i]))
                    mx = std::max(mx, std::fabs(f32[i]));
            if (mx > 0.f)
            {
                int x;
                (void)std::frexp(mx, &x);
                const int e = std::min(std::max(GP_SPLIT_FIXED_EXP + 1 - x, -100), 100);
                scale = std::ldexp(1.0f, e);
                unscale = std::ldexp(1.0f, -e);
            }
        }
        for (int r = 0; r < rows; ++r)
        {
            const int sr = rowmap ? (*rowmap)[r] : r;
            unsigned short *d = &host[(dst_row0 + r) * cols_pad];
            for (int k = 0; k < cols; ++k)
            {
                if (f32)
                {
                    const float w = f32[(size_t)sr * cols + k] * scale;
                    d[k] = f16_rne_bits(w);
                    d[plane + k] = f16_rne_bits(w - f16_bits_to_float(d[k]));
                }
                else if (nbp == 1)
                    d[k] = f16_rne_bits((float)static_cast<const uint8_t *>(tv->data)[(size_t)sr * cols + k] - 128.0f);
                else
                {
                    // q - 32896 (= 256 (qh - 128) + (ql - 128): the constant of the affine map below) as fp16 + exact remainder:
                    // |remainder| <= 16 = 2^-11 of the plane above it, so that a2 x remainder need not be formed (gemm_planes.h)
                    const float pq = (float)static_cast<const uint16_t *>(tv->data)[(size_t)sr * cols + k] - 32896.0f;
                    d[k] = f16_rne_bits(pq);
                    d[plane + k] = f16_rne_bits(pq - f16_bits_to_float(d[k]));
                }
            }
        }
        return unscale;
    };
    auto upload_pmat = [&](PMat &pm, std::vector<unsigned short> &host, int nbp) -> int {
        pm.nbp = nbp;
        weight_bytes += host.size() * sizeof(unsigned short);
        return upload(&pm.p, host);
    };
    // fp32 matrix (kernel layout, padded) -> device; as three bf16 planes when the bf16x3 GEMMs are selected
    auto upload_matrix = [&](float **dst_f32, unsigned short **dst_bx, const std::vector<float> &w) -> int {
        if (!bx)
        {
            weight_bytes += w.size() * sizeof(float);
            return upload(dst_f32, w);
        }
        std::vector<unsigned short> planes(3 * w.size());
        for (size_t i = 0; i < w.size(); ++i)
            split3_host(w[i], planes[i], planes[w.size() + i], planes[2 * w.size() + i]);
        weight_bytes += planes.size() * sizeof(unsigned short);
        return upload(dst_bx, planes);
    };

    const int G = 4 * Hl; // gate rows per direction
    std::vector<float> whh_h[3], bhh_h[3];
    std::vector<unsigned char> whh_qh[3];
    bool whh_all_u8 = keepq;
    for (int tg = 0; tg < 4 && whh_all_u8; ++tg)
        for (int l = 0; l < 3; ++l)
            for (int dir = 0; dir < 2; ++dir)
            {
                const umx_tensor_view *tv = view(tg, "lstm.weight_hh_l" + std::to_string(l) + (dir ? "_reverse" : ""));
                whh_all_u8 = whh_all_u8 && tv && tv->dtype == UMX_DTYPE_U8 && nelems(tv) == (size_t)G * Hl;
            }
    if (whh_all_u8)
        for (int l = 0; l < 3; ++l)
            whh_qh[l].assign((size_t)8 * S * Hl * 64, 0);
    for (int l = 0; l < 3; ++l)
    {
        whh_h[l].assign((size_t)8 * S * Hl * 64, 0.f);
        bhh_h[l].assign((size_t)8 * S * 64, 0.f);
    }
    for (int tg = 0; tg < 4; ++tg)
    {
        TargetBufs &b = tb[tg];
        std::vector<float> v, w;
        // input / output scaling, duplicated per channel like model.cpp:240-290
        if (!get(tg, "input_scale", CROP, v))
            return UMX_ERR_MODEL;
        w.assign(KX, 0.f);
        for (int k = 0; k < NIN; ++k)
            w[k] = v[k % CROP];
        if (int rc = upload(&b.in_scale, w))
            return rc;
        if (!get(tg, "input_mean", CROP, v))
            return UMX_ERR_MODEL;
        w.assign(KX, 0.f);
        for (int k = 0; k < NIN; ++k)
            w[k] = v[k % CROP];
        if (int rc = upload(&b.in_mean, w))
            return rc;
        if (!get(tg, "output_scale", NBINS, v))
            return UMX_ERR_MODEL;
        // fc3's columns: channel c at [c * MAGP, c * MAGP + 2049) (gemm_common.h); model.cpp:240-290 duplicates per channel
        w.assign(NOUT_PAD, 0.f);
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < NBINS; ++k)
                w[c * MAGP + k] = v[k];
        if (int rc = upload(&b.out_scale, w))
            return rc;
        if (!get(tg, "output_mean", NBINS, v))
            return UMX_ERR_MODEL;
        w.assign(NOUT_PAD, 0.f);
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < NBINS; ++k)
                w[c * MAGP + k] = v[k];
        if (int rc = upload(&b.out_mean, w))
            return rc;
        // fc1 (H x 2974) -> (H x KX), zero K padding
        const bool exact_ok = keepq && !(create_flags & UMX_CREATE_U8_DEQUANT); // integers as exact bf16 planes
        if (gemm_planes)
        {
            std::vector<unsigned short> host;
            const umx_tensor_view *tv = view(tg, "fc1.weight");
            if (exact_ok && all_q("fc1.weight", UMX_DTYPE_U8, (size_t)H * NIN))
            {
                fill_planes(host, 1, H, KX, tv, nullptr, H, NIN, nullptr, 0);
                b.fc1_p.s[0] = tv->scale;
                b.fc1_p.o2[0] = tv->offset + 128.0f * tv->scale;
                if (int rc = upload_pmat(b.fc1_p, host, 1))
                    return rc;
            }
            else
            {
                if (!get(tg, "fc1.weight", (size_t)H * NIN, v))
                    return UMX_ERR_MODEL;
                b.fc1_p.s[0] = fill_planes(host, 2, H, KX, nullptr, v.data(), H, NIN, nullptr, 0);
                if (int rc = upload_pmat(b.fc1_p, host, 2))
                    return rc;
            }
        }
        else if (const umx_tensor_view *tv = view(tg, "fc1.weight"); all_q("fc1.weight", UMX_DTYPE_U8, (size_t)H * NIN))
        {
            if (int rc = upload_q(&b.fc1_q.q, tv, H, NIN, H, KX, nullptr, 0, H))
                return rc;
            b.fc1_q.type = BQ_U8;
            b.fc1_q.s[0] = tv->scale;
            b.fc1_q.o[0] = tv->offset;
        }
        else
        {
            if (!get(tg, "fc1.weight", (size_t)H * NIN, v))
                return UMX_ERR_MODEL;
            w.assign((size_t)H * KX, 0.f);
            for (int o = 0; o < H; ++o)
                memcpy(&w[(size_t)o * KX], &v[(size_t)o * NIN], sizeof(float) * NIN);
            if (int rc = upload_matrix(&b.fc1_w, &b.fc1_bx, w))
                return rc;
        }
        const char *bnn[4] = {"running_mean", "running_var", "weight", "bias"};
        for (int k = 0; k < 4; ++k)
        {
            if (!get(tg, std::string("bn1.") + bnn[k], H, v))
                return UMX_ERR_MODEL;
            if (int rc = upload(&b.bn1[k], v))
                return rc;
            if (!get(tg, std::string("bn2.") + bnn[k], H, v))
                return UMX_ERR_MODEL;
            if (int rc = upload(&b.bn2[k], v))
                return rc;
            if (!get(tg, std::string("bn3.") + bnn[k], NOUT, v))
                return UMX_ERR_MODEL;
            w.assign(NOUT_PAD, k == 1 ? 1.f : 0.f); // padded running_var = 1: no 0/0 in dead columns
            for (int c = 0; c < 2; ++c)
                memcpy(&w[c * MAGP], &v[c * NBINS], sizeof(float) * NBINS);
            if (int rc = upload(&b.bn3[k], w))
                return rc;
        }
        if (gemm_planes)
        {
            std::vector<unsigned short> host;
            const umx_tensor_view *tv = view(tg, "fc2.weight");
            if (exact_ok && all_q("fc2.weight", UMX_DTYPE_U16, (size_t)H * 2 * H))
            {
                fill_planes(host, 2, H, 2 * H, tv, nullptr, H, 2 * H, nullptr, 0);
                b.fc2_p.s[0] = tv->scale;
                b.fc2_p.o2[0] = tv->offset + 32896.0f * tv->scale;
                if (int rc = upload_pmat(b.fc2_p, host, 2))
                    return rc;
            }
            else
            {
                if (!get(tg, "fc2.weight", (size_t)H * 2 * H, v))
                    return UMX_ERR_MODEL;
                b.fc2_p.s[0] = fill_planes(host, 2, H, 2 * H, nullptr, v.data(), H, 2 * H, nullptr, 0);
                if (int rc = upload_pmat(b.fc2_p, host, 2))
                    return rc;
            }
        }
        else if (const umx_tensor_view *tv = view(tg, "fc2.weight"); all_q("fc2.weight", UMX_DTYPE_U16, (size_t)H * 2 * H))
        {
            if (int rc = upload_q(&b.fc2_q.q, tv, H, 2 * H, H, 2 * H, nullptr, 0, H))
                return rc;
            b.fc2_q.type = BQ_U16;
            b.fc2_q.s[0] = tv->scale;
            b.fc2_q.o[0] = tv->offset;
        }
        else
        {
            if (!get(tg, "fc2.weight", (size_t)H * 2 * H, v))
                return UMX_ERR_MODEL;
            if (int rc = upload_matrix(&b.fc2_w, &b.fc2_bx, v))
                return rc;
        }
        // fc3's output rows in the column layout of the mask planes: channel c's 2049 rows at [c * MAGP, ...), zero rows between
        std::vector<unsigned char> fc3_perm;
        umx_tensor_view fc3_tv;
        memset(&fc3_tv, 0, sizeof fc3_tv);
        if (const umx_tensor_view *tv = view(tg, "fc3.weight"); tv && tv->dtype == UMX_DTYPE_U16 && nelems(tv) == (size_t)NOUT * H)
        {
            fc3_perm.assign((size_t)NOUT_PAD * H * 2, 0);
            for (int c = 0; c < 2; ++c)
                memcpy(&fc3_perm[(size_t)c * MAGP * H * 2], static_cast<const unsigned char *>(tv->data) + (size_t)c * NBINS * H * 2, (size_t)NBINS * H * 2);
            fc3_tv = *tv;
            fc3_tv.data = fc3_perm.data();
        }
        auto fc3_f32 = [&](std::vector<float> &dst) -> bool { // dequantised fp32, permuted, (NOUT_PAD x H)
            std::vector<float> src;
            if (!get(tg, "fc3.weight", (size_t)NOUT * H, src))
                return false;
            dst.assign((size_t)NOUT_PAD * H, 0.f);
            for (int c = 0; c < 2; ++c)
                memcpy(&dst[(size_t)c * MAGP * H], &src[(size_t)c * NBINS * H], sizeof(float) * (size_t)NBINS * H);
            return true;
        };
        if (gemm_planes)
        {
            std::vector<unsigned short> host;
            const umx_tensor_view *tv = fc3_tv.data ? &fc3_tv : nullptr;
            if (exact_ok && all_q("fc3.weight", UMX_DTYPE_U16, (size_t)NOUT * H))
            {
                fill_planes(host, 2, NOUT_PAD, H, tv, nullptr, NOUT_PAD, H, nullptr, 0);
                b.fc3_p.s[0] = tv->scale;
                b.fc3_p.o2[0] = tv->offset + 32896.0f * tv->scale;
                if (int rc = upload_pmat(b.fc3_p, host, 2))
                    return rc;
            }
            else
            {
                if (!fc3_f32(v))
                    return UMX_ERR_MODEL;
                b.fc3_p.s[0] = fill_planes(host, 2, NOUT_PAD, H, nullptr, v.data(), NOUT_PAD, H, nullptr, 0);
                if (int rc = upload_pmat(b.fc3_p, host, 2))
                    return rc;
            }
        }
        else if (const umx_tensor_view *tv = &fc3_tv; all_q("fc3.weight", UMX_DTYPE_U16, (size_t)NOUT * H))
        {
            if (int rc = upload_q(&b.fc3_q.q, tv, NOUT_PAD, H, NOUT_PAD, H, nullptr, 0, NOUT_PAD))
                return rc;
            b.fc3_q.type = BQ_U16;
            b.fc3_q.s[0] = tv->scale;
            b.fc3_q.o[0] = tv->offset;
        }
        else
        {
            if (!fc3_f32(w))
                return UMX_ERR_MODEL;
            if (int rc = upload_matrix(&b.fc3_w, &b.fc3_bx, w))
                return rc;
        }
        // LSTM: permute gate rows so a workgroup's 64 columns (g,u) are contiguous
        for (int l = 0; l < 3; ++l)
        {
            const umx_tensor_view *ihv[2] = {view(tg, "lstm.weight_ih_l" + std::to_string(l)),
                                             view(tg, "lstm.weight_ih_l" + std::to_string(l) + "_reverse")};
            const bool ih_q8 = all_q("lstm.weight_ih_l" + std::to_string(l), UMX_DTYPE_U8, (size_t)G * H) &&
                               all_q("lstm.weight_ih_l" + std::to_string(l) + "_reverse", UMX_DTYPE_U8, (size_t)G * H);
            const bool ih_exact = gemm_planes && exact_ok && ih_q8;
            const bool ih_q = ih_q8 && (!gemm_planes || ih_exact); // the source stays u8 (no fp32 copy needed)
            std::vector<unsigned short> ih_planes;
            std::vector<float> ihw((ih_q || gemm_planes) ? 0 : (size_t)2 * G * H), ihb((size_t)2 * G);
            for (int dir = 0; dir < 2; ++dir)
            {
                const std::string sfx = "_l" + std::to_string(l) + (dir ? "_reverse" : "");
                std::vector<float> wih, whhv, bih, bhhv;
                if ((!ih_q && !get(tg, "lstm.weight_ih" + sfx, (size_t)G * H, wih)) ||
                    (!whh_all_u8 && !get(tg, "lstm.weight_hh" + sfx, (size_t)G * Hl, whhv)) ||
                    !get(tg, "lstm.bias_ih" + sfx, G, bih) || !get(tg, "lstm.bias_hh" + sfx, G, bhhv))
                    return UMX_ERR_MODEL;
                const int chain = tg * 2 + dir;
                const umx_tensor_view *hhv = view(tg, "lstm.weight_hh" + sfx);
                const unsigned char *hhq = whh_all_u8 ? static_cast<const unsigned char *>(hhv->data) : nullptr;
                if (whh_all_u8)
                {
                    whh_s[l][chain] = hhv->scale;
                    whh_o[l][chain] = hhv->offset;
                }
                std::vector<int> rowmap(G); // destination gate-interleaved row -> PyTorch gate row
                for (int sl = 0; sl < S; ++sl)
                    for (int g = 0; g < 4; ++g)
                        for (int u = 0; u < 16; ++u)
                        {
                            const int row = g * Hl + sl * 16 + u; // PyTorch gate row (i|f|g|o blocks)
                            const int col = u * 4 + g; // the 4 gates of a unit share a DPP quad
                            const size_t n = (size_t)dir * G + (size_t)sl * 64 + col;
                            rowmap[sl * 64 + col] = row;
                            if (!ih_q && !gemm_planes)
                                memcpy(&ihw[n * H], &wih[(size_t)row * H], sizeof(float) * H);
                            ihb[n] = bih[row];
                            bhh_h[l][((size_t)chain * S + sl) * 64 + col] = bhhv[row];
                            for (int k = 0; k < Hl; ++k)
                            {
                                const size_t di = (((size_t)chain * S + sl) * Hl + k) * 64 + col;
                                if (whh_all_u8)
                                    whh_qh[l][di] = hhq[(size_t)row * Hl + k];
                                else
                                    whh_h[l][di] = whhv[(size_t)row * Hl + k];
                            }
                        }
                if (gemm_planes)
                {
                    if (ih_exact)
                    {
                        fill_planes(ih_planes, 1, (size_t)2 * G, H, ihv[dir], nullptr, G, H, &rowmap, (size_t)dir * G);
                        b.ih_p[l].s[dir] = ihv[dir]->scale;
                        b.ih_p[l].o2[dir] = ihv[dir]->offset + 128.0f * ihv[dir]->scale;
                    }
                    else
                        b.ih_p[l].s[dir] = fill_planes(ih_planes, 2, (size_t)2 * G, H, nullptr, wih.data(), G, H, &rowmap, (size_t)dir * G);
                }
                else if (ih_q)
                {
                    if (int rc = upload_q(&b.ih_q[l].q, ihv[dir], G, H, G, H, &rowmap, (size_t)dir * G, (size_t)2 * G))
                        return rc;
                    b.ih_q[l].type = BQ_U8;
                    b.ih_q[l].s[dir] = ihv[dir]->scale;
                    b.ih_q[l].o[dir] = ihv[dir]->offset;
                }
            }
            if (gemm_planes)
            {
                if (int rc = upload_pmat(b.ih_p[l], ih_planes, ih_exact ? 1 : 2))
                    return rc;
            }
            else if (!ih_q)
            {
                if (int rc = upload_matrix(&b.ih_w[l], &b.ih_bx[l], ihw))
                    return rc;
            }
            if (int rc = upload(&b.ih_b[l], ihb))
                return rc;
        }
    }
    for (int l = 0; l < 3; ++l)
    {
        if (whh_all_u8)
        {
            if (int rc = upload(&whh_q[l], whh_qh[l]))
                return rc;
            weight_bytes += whh_qh[l].size();
        }
        else
        {
            if (int rc = upload(&whh[l], whh_h[l]))
                return rc;
            weight_bytes += whh_h[l].size() * sizeof(float);
        }
        if (int rc = upload(&bhh[l], bhh_h[l]))
            return rc;
    }
    // ---- tables: window (dsp.hpp:61-78, the reference's float PI), window sum-square
    // (dsp.hpp:80-101, same accumulation order), FFT twiddles (rounded from double)
    {
        std::vector<float> w(NFFT);
        static const float PI = 3.14159265359F;
        const float floatN = (float)(NFFT + 1);
        for (int n = 0; n < NFFT; ++n)
            w[n] = 0.5F * (1.0F - cosf(2.0F * PI * (float)n / (floatN - 1)));
        if (int rc = upload(&window, w))
            return rc;
        const size_t total = (size_t)NFFT + (size_t)HOP * (T - 1);
        std::vector<float> nwh(total, 0.f);
        for (int i = 0; i < T; ++i)
        {
            const size_t s0 = (size_t)i * HOP;
            for (size_t j = s0; j < std::min(total, s0 + NFFT); ++j)
                nwh[j] += w[j - s0] * w[j - s0];
        }
        if (int rc = upload(&nw, nwh))
            return rc;
        std::vector<float2> t1(256), t2(4096);
        for (int r = 0; r < 16; ++r)
            for (int k = 0; k < 16; ++k)
            {
                const double ph = -2.0 * M_PI * (double)(r * k) / 256.0;
                t1[r * 16 + k] = make_float2((float)cos(ph), (float)sin(ph));
            }
        for (int r = 0; r < 16; ++r)
            for (int j = 0; j < 256; ++j)
            {
                const double ph = -2.0 * M_PI * (double)(r * j) / 4096.0;
                t2[r * 256 + j] = make_float2((float)cos(ph), (float)sin(ph));
            }
        if (int rc = upload(&tw1, t1))
            return rc;
        if (int rc = upload(&tw2, t2))
            return rc;
    }
    // ---- per-segment buffers: two pipeline slots x B track lanes
    if (int rc = dalloc(&audio_in, (size_t)2 * N))
        return rc;
    for (int k = 0; k < 4; ++k)
        if (int rc = dalloc(&out_dev[k], (size_t)2 * N))
            return rc;
    if (int rc = dalloc(&state, state_floats() * B))
        return rc;
    if (int rc = dalloc(&state_alt, state_floats() * B))
        return rc;
    if (int rc = dalloc(&backup, (size_t)kBackupCalls * 3 * state_floats() * B))
        return rc;
    lsync_words = LSTM_SYNC_HEADER_WORDS + std::max(granule_count(S) * 2, lstm_batched ? lstmb_granule_words(Hl) * ((B + LSTMB_GROUP_TRACKS - 1) / LSTMB_GROUP_TRACKS) : (size_t)0);
    // Two slots.  (Three were tried for single-track contexts in round 3 -- a third segment in flight has its front stage
    // done by the time an LSTM grid retires, so that two grids would be resident all the time: 7.57 ms per segment against
    // 6.70 with two; the grids and the GEMM blocks beside them only slow each other down, avg LSTM launch 3.03 -> 3.79 ms.)
    nslots = 2;
    if (gemm_planes)
    {
        // the plane GEMMs address their operands and fc3's mask output through buffer resources with 32-bit byte offsets
        // (gemm_planes.h: both planes of an operand behind one base; gemm_common.h: all lanes' masks behind one base): refuse
        // lane x segment-length combinations that do not fit instead of reading zeros past the range check
        const unsigned long long rows = (unsigned long long)B * Tp + Mpad;
        const unsigned long long planes_bytes = 2ull * rows * (unsigned long long)std::max(KX, 2 * H) * 2ull;
        const unsigned long long mask_bytes = (unsigned long long)B * 2ull * T * MAGP * 4ull;
        if (planes_bytes >= (1ull << 31) || mask_bytes >= (1ull << 32))
        {
            set_error("track lanes x segment length exceed the 32-bit addressing of the plane GEMMs' operands: fewer lanes or a shorter segment");
            return UMX_ERR_ARG;
        }
    }
    for (int si = 0; si < nslots; ++si)
    {
        Slot &sl = slot[si];
        UMX_HIP_CHECK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        // Buffers that a launch covering several track lanes reads or writes (the batched LSTM kernel, the plane GEMMs
        // with M = lanes x Tp) are ONE allocation per slot (and target), lane after lane at a constant stride.
        float *x_all = nullptr;
        if (int rc = dalloc(&x_all, ((size_t)B * Tp + Mpad) * KX))
            return rc;
        for (int tg = 0; tg < 4; ++tg)
        {
            float *cat_all = nullptr, *la_all = nullptr, *lb_all = nullptr, *P_all = nullptr, *a2_all = nullptr, *mag_all = nullptr;
            if (int rc = dalloc(&cat_all, ((size_t)B * Tp + Mpad) * 2 * H))
                return rc;
            if (int rc = dalloc(&la_all, ((size_t)B * Tp + Mpad) * H))
                return rc;
            if (int rc = dalloc(&lb_all, ((size_t)B * Tp + Mpad) * H))
                return rc;
            if (int rc = dalloc(&P_all, ((size_t)B * Tp + Mpad) * 4 * H))
                return rc;
            if (int rc = dalloc(&a2_all, ((size_t)B * Tp + Mpad) * H))
                return rc;
            if (int rc = dalloc(&mag_all, (size_t)B * 2 * T * MAGP))
                return rc;
            unsigned short *xs_p = nullptr, *cat_p = nullptr, *la_p = nullptr, *lb_p = nullptr, *a2_p = nullptr;
            float *rs[8] = {};
            if (gemm_planes) // planes [2][B * Tp][K]: plane-major over ALL lanes, so that M runs across the lanes
            {
                if (int rc = dalloc(&xs_p, 2 * ((size_t)B * Tp + Mpad) * KX))
                    return rc;
                if (int rc = dalloc(&cat_p, 2 * ((size_t)B * Tp + Mpad) * 2 * H))
                    return rc;
                for (unsigned short **q : {&la_p, &lb_p, &a2_p})
                    if (int rc = dalloc(q, 2 * ((size_t)B * Tp + Mpad) * H))
                        return rc;
                for (int k = 0; k < 8; ++k) // [2..4]: the recurrence's outputs, one array per direction when it writes the planes itself
                    if (int rc = dalloc(&rs[k], ((size_t)B * Tp + Mpad) * (k >= 2 && k <= 4 ? 2 : 1)))
                        return rc;
            }
            for (int ln = 0; ln < B; ++ln)
            {
                TargetAct &b = sl.lane[ln].ta[tg];
                b.cat = cat_all + (size_t)ln * Tp * 2 * H;
                b.la = la_all + (size_t)ln * Tp * H;
                b.lb = lb_all + (size_t)ln * Tp * H;
                b.P = P_all + (size_t)ln * Tp * 4 * H;
                b.a2 = a2_all + (size_t)ln * Tp * H;
                b.mag = mag_all + (size_t)ln * 2 * T * MAGP;
                if (gemm_planes) // lane ln's rows start at row ln * Tp of every plane
                {
                    b.xs_p = xs_p + (size_t)ln * Tp * KX;
                    b.cat_p = cat_p + (size_t)ln * Tp * 2 * H;
                    b.la_p = la_p + (size_t)ln * Tp * H;
                    b.lb_p = lb_p + (size_t)ln * Tp * H;
                    b.a2_p = a2_p + (size_t)ln * Tp * H;
                    b.rs_xs = rs[0] + (size_t)ln * Tp;
                    b.rs_catL = rs[1] + (size_t)ln * Tp;
                    b.rs_catR = rs[2] + (size_t)ln * Tp;
                    b.rs_la = rs[3] + (size_t)ln * Tp;
                    b.rs_lb = rs[4] + (size_t)ln * Tp;
                    b.rs_a2 = rs[5] + (size_t)ln * Tp;
                    b.rsc_xs = rs[6] + (size_t)ln * Tp;
                    b.rsc_a2 = rs[7] + (size_t)ln * Tp;
                }
            }
        }
        {
            // every lane's streaming buffers a fixed stride apart (lane_strides()): the streaming kernels take lane 0's
            // pointers and cover all active lanes in one launch (common.h LaneSet)
            const WienerStrides ls = lane_strides();
            float2 *spec_all, *y_all, *frames_all;
            float *wpart_all, *Rc_all;
            unsigned *maxabs_all;
            if (int rc = dalloc(&spec_all, (size_t)B * ls.spec))
                return rc;
            if (int rc = dalloc(&y_all, (size_t)B * ls.y))
                return rc;
            if (int rc = dalloc(&frames_all, (size_t)B * ls.frames))
                return rc;
            if (int rc = dalloc(&wpart_all, (size_t)B * ls.part))
                return rc;
            if (int rc = dalloc(&Rc_all, (size_t)B * ls.rc))
                return rc;
            if (int rc = dalloc(&maxabs_all, (size_t)B))
                return rc;
            for (int ln = 0; ln < B; ++ln)
            {
                Lane &L = sl.lane[ln];
                L.x = x_all + (size_t)ln * Tp * KX;
                L.spec = spec_all + (size_t)ln * ls.spec;
                L.y = y_all + (size_t)ln * ls.y;
                L.frames = frames_all + (size_t)ln * ls.frames;
                L.wpart = wpart_all + (size_t)ln * ls.part;
                L.Rc = Rc_all + (size_t)ln * ls.rc;
                L.maxabs = maxabs_all + ln;
            }
        }
        if (int rc = dalloc(&sl.status, 4))
            return rc;
        if (int rc = dalloc(&sl.hbuf, (size_t)2 * 8 * Hl))
            return rc;
        if (int rc = dalloc(&sl.lsync, lsync_words))
            return rc;
        if (int rc = dalloc(&sl.lprof, 1024 + 64 * 8 * 5))
            return rc;
        for (int i = 0; i <= ST_COUNT; ++i)
            UMX_HIP_CHECK(hipEventCreate(&sl.ev[i]));
        for (int i : {ST_FC1, ST_IH0, ST_IH1, ST_IH2, ST_FC2, ST_FC3})
            UMX_HIP_CHECK(hipEventCreate(&sl.evk[i]));
        for (int l = 0; l < 3; ++l)
            UMX_HIP_CHECK(hipEventCreateWithFlags(&sl.rec_done[l], hipEventDisableTiming));
    }
    stream = slot[0].stream;
    {
        // residency of the persistent LSTM kernel: the smaller of the two activation flavours of the
        // instantiation this hidden size uses (the two-grid co-residency of the pipeline rests on it)
        const int kpw = Hl / 8;
        int per_cu = 1 << 30, cus = 0;
        for (int precise = 0; precise < 2; ++precise)
        {
            const void *fn = lstm_persistent_fn(kpw, precise != 0);
            int v = 0;
            UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, lstm_threads, 0));
            per_cu = std::min(per_cu, v);
        }
        UMX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        n_cus = cus;
        lstm_capacity = per_cu * cus;
        if (lstm_batched)
        {
            // the batched kernel: worst-case dynamic LDS (16 lanes), both activation flavours
            const size_t lds_max = lstmb_lds_bytes(LSTMB_GROUP_TRACKS, 8);
            per_cu = 1 << 30;
            for (int precise = 0; precise < 2; ++precise)
                for (int wq = 0; wq < 2; ++wq)
                {
                    const void *fn = lstm_batch_fn(Hl, wq != 0, precise != 0);
                    if (!fn)
                    {
                        set_error("track batching needs hidden_size in {128, 256, 512, 1024}");
                        return UMX_ERR_ARG;
                    }
                    UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
                    int v = 0;
                    UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, LSTM_THREADS, lstmb_lds_bytes(B > 8 ? 16 : B > 4 ? 8 : B > 2 ? 4 : B > 1 ? 2 : 1, B > 8 ? 8 : 16)));
                    per_cu = std::min(per_cu, v);
                }
            if (B > LSTMB_GROUP_TRACKS) // more than 16 lanes: the two-group kernel (u8-resident W_hh only)
            {
                for (int l = 0; l < 3; ++l)
                    if (!whh_q[l] || u8_dequant)
                    {
                        set_error("more than 16 track lanes need the u8-resident W_hh (quantised model, no UMX_CREATE_U8_DEQUANT / _DEQUANTISE_AT_LOAD)");
                        return UMX_ERR_ARG;
                    }
                for (int groups = 2; groups <= 3; ++groups)
                    for (int precise = 0; precise < 2; ++precise)
                    {
                        const void *fn = lstm_batch2_fn(Hl, groups, precise != 0);
                        const size_t l2 = lstmb2_lds_bytes(groups, lstmb2_bulk(groups));
                        UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
                        int v = 0;
                        UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, LSTMB2_THREADS, l2));
                        per_cu = std::min(per_cu, v);
                    }
                // ... or two groups side by side, each chain 16 workgroups of two slices: one workgroup per CU
                lstm_batchs_ok = false;
                if (lstm_batchs_fn(Hl, 2, false) && S % kBatchsSpan == 0)
                {
                    const size_t lg = lstmb_lds_bytes(LSTMB_GROUP_TRACKS, kBatchsBulk, kBatchsSpan);
                    lstm_batchs_ok = true;
                    for (int precise = 0; precise < 2; ++precise)
                    {
                        const void *fn = lstm_batchs_fn(Hl, 2, precise != 0);
                        UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lg));
                        int v = 0;
                        UMX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, LSTM_THREADS, lg));
                        lstm_batchs_ok = lstm_batchs_ok && v >= 1 && 2 * 8 * (S / kBatchsSpan) <= v * cus;
                    }
                }
            }
            lstm_batch_capacity = per_cu * cus;
            // the recurrence writes the plane GEMMs' A operands (layers 1, 2 and fc2's right half) and their row sums itself where
            // it runs on the u8-resident W_hh (its gate lanes hold h as two fp16 planes, its all-ones tile the row sums): no
            // split_planes launches for them (lstm_batch.h, LstmBArgs::planes).  -DUMX_FUSE_LSTM_PLANES=0: A/B builds
            // More than 32 lanes (lstm_batch2.h, no register left): only the row sums; split_planes_kernel still writes the planes -- the
            // same bits, so a track's result does not depend on the size of the context.
            lstm_rowsums = UMX_FUSE_LSTM_PLANES && gemm_planes && !u8_dequant && whh_q[0] && whh_q[1] && whh_q[2];
            lstm_writes_planes = lstm_rowsums && B <= 2 * LSTMB_GROUP_TRACKS;
        }
    }
    // dynamic LDS > 64 KiB must be opted into
    {
        const void *bxs[10] = {reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC1, BQ_U8X>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_IH, BQ_U8X>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC1, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC1, BQ_U8>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_IH, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_IH, BQ_U8>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC2, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC2, BQ_U16>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC3, BQ_F32>),
                              reinterpret_cast<const void *>(gemm_bf16x3_kernel<G_FC3, BQ_U16>)};
        for (const void *fn : bxs)
            UMX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, BX_LDS_BYTES));
        {
            const int wi_lds = 4 * FFT_LDS_ELEMS * (int)sizeof(float2); // 139,264 (NSRC = 4; 2 and 1 need less)
            UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wiener_istft_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, wi_lds));
            UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wiener_istft_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, wi_lds));
        }
#define UMX_GP_ATTR(MODE)                                                                                              \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 1, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(2, 2, 1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(2, 2, 2))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 1, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_kernel<MODE, 2, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 2))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_pp_kernel<MODE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 1))); \
    UMX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_planes_pp_kernel<MODE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, gp_lds_bytes(4, 4, 2)));
        UMX_GP_ATTR(G_FC1)
        UMX_GP_ATTR(G_IH)
        UMX_GP_ATTR(G_FC2)
        UMX_GP_ATTR(G_FC3)
#undef UMX_GP_ATTR
    }
    UMX_HIP_CHECK(hipDeviceSynchronize());
    return UMX_OK;
}

int umx_hip_ctx::ensure_staging()
{
    if (stage_in[0])
        return UMX_OK;
    UMX_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int si = 0; si < nslots; ++si)
    {
        for (hipEvent_t *e : {&slot[si].k_done, &slot[si].out_free})
            UMX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        if (int rc = dalloc(&stage_in[si], (size_t)2 * N * B, false))
            return rc;
        for (int k = 0; k < 4 * B; ++k)
            if (int rc = dalloc(&stage_out[si][k], (size_t)2 * N, false))
                return rc;
    }
    return UMX_OK;
}

// Host-pointer calls: where the stems of call k go out (round 3; kernel + copy timelines by tools/pcie_trace.sh).
//   * On the slot's own stream right behind its kernels (rounds 1-2), the download (48 ms for 32 lanes) stands in front of
//     call k + 2's upload and kernels; the two slots then fall into lock step -- kernels of two calls, downloads of two
//     calls, uploads of two calls, nothing overlapping: 128 ms per step against 78 ms of kernels.
//   * On the slot's own stream one call LATER (behind call k + 2's front stage): the same 129 ms.
//   * On the OTHER slot's stream behind call k + 1's kernels: 110-113 ms (the copies stay on the DMA engines, but a call's
//     kernels end at about the same time as the next call's, so the download still starts late).
//   * On a copy stream of its own behind an event (this code): 91-93 ms.  The runtime executes these copies as shader blits
//     (a download that does not follow kernels of its own stream), which wait for compute units behind the persistent LSTM
//     grids; a marker kernel in front of them does not change that.
// A separate UPLOAD stream as well made everything serial (141 ms): streams beyond the runtime's hardware queues share one.
int umx_hip_ctx::queue_download(const DeferredDownload &d, hipStream_t on)
{
    Slot &src = slot[d.si];
    if (on != src.stream)
        UMX_HIP_CHECK(hipStreamWaitEvent(on, src.k_done, 0));
    for (int ln = 0; ln < d.nb; ++ln)
        if (d.n[ln] > 0)
            for (int s2 = 0; s2 < 4; ++s2)
                UMX_HIP_CHECK(hipMemcpyAsync(d.host[4 * ln + s2], stage_out[d.si][4 * ln + s2], sizeof(float) * 2 * (size_t)d.n[ln],
                                             hipMemcpyDeviceToHost, on));
    UMX_HIP_CHECK(hipEventRecord(src.out_free, on));
    src.out_free_valid = true;
    return UMX_OK;
}

int umx_hip_ctx::sync_all()
{
    for (int si = 0; si < nslots; ++si)
        UMX_HIP_CHECK(hipStreamSynchronize(slot[si].stream));
    if (copy_stream)
        UMX_HIP_CHECK(hipStreamSynchronize(copy_stream));
    for (int si = 0; si < nslots; ++si)
        slot[si].out_free_valid = false; // drained
    return UMX_OK;
}

// ---------------------------------------------------------------- LSTM layer
int umx_hip_ctx::run_lstm_layer(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask)
{
    if (lstm_batched)
        return run_lstm_layer_batched(sl, layer, active, nact, stepwise, lane_mask);
    hipStream_t st = sl.stream;
    LstmArgs a;
    memset(&a, 0, sizeof a);
    a.W = whh[layer];      // nullptr when W_hh is u8-resident
    a.Wq = whh_q[layer];
    for (int c = 0; c < 8; ++c)
    {
        a.wsc[c] = whh_s[layer][c];
        a.wof[c] = whh_o[layer][c];
    }
    a.bhh = bhh[layer];
    a.state = state;
    a.hbuf = sl.hbuf;
    a.sync = sl.lsync;
    a.status = sl.status;
    a.prof = (last_flags & UMX_FLAG_LSTM_PROFILE) ? sl.lprof : nullptr;
    a.force_safe = (last_flags & UMX_FLAG_LSTM_FORCE_SAFE) ? 1 : 0;
    a.poll_delay = lstm_poll_delay;
    a.Hl = Hl;
    a.S = S;
    a.T = T;
    a.ldp = 4 * H;
    a.layer = layer;
    for (int i = 0; i < 4; ++i)
    {
        const TargetAct &b = sl.lane[0].ta[i];
        a.P[i] = b.P;
        if (layer == 0)
        {
            a.out[i] = b.la;
            a.ldo = H;
            a.col0 = 0;
        }
        else if (layer == 1)
        {
            a.out[i] = b.lb;
            a.ldo = H;
            a.col0 = 0;
        }
        else
        {
            a.out[i] = b.cat; // inference.cpp:118-123 skip concat: lstm output -> right half of cat
            a.ldo = 2 * H;
            a.col0 = H;
        }
        a.tmap[i] = i < nact ? active[i] : 0;
    }
    const int nchains = 2 * nact;
    a.nchains = nchains;
    const dim3 grid(S, nchains), block(LSTM_THREADS);
    const int kpw = Hl / 8;
    bool persistent = !stepwise && persistent_ok && (kpw == 8 || kpw == 16 || kpw == 32 || kpw == 64) &&
                      8 * S <= lstm_capacity;
    if (persistent)
    {
        a.tag_base = next_tag_base();
        a.abort_at = ((last_flags & UMX_FLAG_DEBUG_LSTM_ABORT) && layer == 1) ? T / 2 : 0;
        // census + arrival counter every launch; the granule area only when the tag epoch wraps
        UMX_HIP_CHECK(hipMemsetAsync(sl.lsync, 0, sizeof(unsigned) * (tag_epoch == 0 ? lsync_words : LSTM_SYNC_HEADER_WORDS), st));
        void *kargs[] = {&a};
        const void *fn = lstm_persistent_fn(kpw, last_flags & UMX_FLAG_PRECISE_ACT);
        // The granule exchange needs the whole grid co-resident.  Residency was checked against the
        // occupancy of this kernel at create time (lstm_capacity); a plain launch is used because ROCm
        // serialises cooperative launches against other queues, which would defeat the two-slot overlap.
        // Every spin in the kernel is bounded, so a grid that is not resident after all ends as
        // UMX_ERR_TIMEOUT, not as a hang.
        // always 8*S workgroups: with round-robin dispatch every XCD then receives S of them and the
        // census can enable the intra-XCD protocol; surplus workgroups (skipped targets) exit at once
        hipError_t e = lstm_gate_launch(device, st, 8 * S * (lstm_threads > 512 ? 2 : 1), 2 * n_cus,
                                        [&] { return hipLaunchKernel(fn, dim3(8 * S), dim3(lstm_threads), kargs, 0, st); });
        if (e != hipSuccess)
        {
            (void)hipGetLastError();
            persistent_ok = false;
            persistent = false;
        }
    }
    if (!persistent)
    {
        hipLaunchKernelGGL(lstm_state_to_hbuf, dim3(nchains), dim3(256), 0, st, a, nchains);
        if (last_flags & UMX_FLAG_PRECISE_ACT)
            for (int step = 0; step < T; ++step)
                hipLaunchKernelGGL(lstm_step_kernel<true>, grid, block, 0, st, a, step);
        else
            for (int step = 0; step < T; ++step)
                hipLaunchKernelGGL(lstm_step_kernel<false>, grid, block, 0, st, a, step);
        hipLaunchKernelGGL(lstm_hbuf_to_state, dim3(nchains), dim3(256), 0, st, a, nchains);
    }
    sl.last_persistent = persistent;
    UMX_HIP_CHECK(hipGetLastError());
    return UMX_OK;
}

// All track lanes of `lane_mask` through ONE launch per layer (lstm_batch.h).  stepwise (or a grid that cannot be
// co-resident): the same kernel one step per launch, carried through the fp32 stream state -- bit-identical.
int umx_hip_ctx::run_lstm_layer_batched(Slot &sl, int layer, const int *active, int nact, bool stepwise, unsigned long long lane_mask)
{
    hipStream_t st = sl.stream;
    LstmBArgs a;
    memset(&a, 0, sizeof a);
    a.W = whh[layer];
    a.Wq = whh_q[layer];
    for (int c = 0; c < 8; ++c)
    {
        a.wsc[c] = whh_s[layer][c];
        a.wof[c] = whh_o[layer][c];
    }
    a.bhh = bhh[layer];
    a.state = state;
    a.state_out = state;
    a.state_stride = state_floats();
    a.sync = sl.lsync;
    a.status = sl.status;
    a.prof = (last_flags & UMX_FLAG_LSTM_PROFILE) ? sl.lprof : nullptr;
    a.force_safe = (last_flags & UMX_FLAG_LSTM_FORCE_SAFE) ? 1 : 0;
    a.Hl = Hl;
    a.S = S;
    a.T = T;
    a.ldp = 4 * H;
    a.layer = layer;
    a.p_stride = (size_t)Tp * 4 * H;
    a.out_stride = layer == 2 ? (size_t)Tp * 2 * H : (size_t)Tp * H;
    a.ldo = layer == 2 ? 2 * H : H;
    a.col0 = layer == 2 ? H : 0; // inference.cpp:118-123 skip concat: lstm output -> right half of cat
    for (int i = 0; i < 4; ++i)
    {
        const TargetAct &b = sl.lane[0].ta[i];
        a.P[i] = b.P;
        a.out[i] = layer == 0 ? b.la : layer == 1 ? b.lb : b.cat;
        a.tmap[i] = i < nact ? active[i] : 0;
    }
    a.nchains = 2 * nact;
    a.lane_mask = lane_mask;
    const bool wq_layer = whh_q[layer] != nullptr && !u8_dequant;
    const bool fuse = lstm_rowsums && wq_layer;
    if (fuse)
    {
        const size_t rows_all = (size_t)B * Tp + Mpad;
        for (int i = 0; i < 4; ++i)
        {
            const TargetAct &b = sl.lane[0].ta[i];
            a.planes[i] = layer == 0 ? b.la_p : layer == 1 ? b.lb_p : b.cat_p; // (dropped below if the launch's kernel cannot write them)
            a.rs_dir[i] = layer == 0 ? b.rs_la : layer == 1 ? b.rs_lb : b.rs_catR;
        }
        a.ldpl = a.ldo;
        a.plane_elems = rows_all * (size_t)a.ldo;
        a.rs_rows = rows_all;
    }
    a.Tp = Tp;
    int top = 0;
    for (int ln = 0; ln < LSTMB_MAX_TRACKS; ++ln)
        if ((lane_mask >> ln) & 1ull)
            top = ln + 1;
    const int groups = (top + LSTMB_GROUP_TRACKS - 1) / LSTMB_GROUP_TRACKS; // > 1: lstm_batch2.h, groups of 16 lanes in turn
    a.nbp = top > 8 ? 16 : top > 4 ? 8 : top > 2 ? 4 : top > 1 ? 2 : 1;
    const bool wq = whh_q[layer] != nullptr && !u8_dequant;
    // 17 .. 32 lanes: the two groups side by side on the chip (lstm_batchs_kernel: half the hand-off bytes) where it fits, else -- and
    // for 33 .. 48 lanes -- the groups in turn through one twelve-wave workgroup (lstm_batch2.h).  Same bits either way.
    const char *ge = getenv("UMX_LSTM_GROUPED"); // 0: always the groups in turn; read per launch: the tests switch it
    const bool grouped = groups == 2 && lstm_batchs_ok && !(ge && atoi(ge) == 0);
    a.bulk = grouped ? kBatchsBulk : groups > 1 ? lstmb2_bulk(groups) : a.nbp > 8 ? 8 : 16;
    const size_t lds = grouped ? lstmb_lds_bytes(LSTMB_GROUP_TRACKS, a.bulk, kBatchsSpan) : groups > 1 ? lstmb2_lds_bytes(groups, a.bulk) : lstmb_lds_bytes(a.nbp, a.bulk);
    const void *fn = grouped      ? lstm_batchs_fn(Hl, groups, last_flags & UMX_FLAG_PRECISE_ACT)
                     : groups > 1 ? lstm_batch2_fn(Hl, groups, last_flags & UMX_FLAG_PRECISE_ACT)
                                  : lstm_batch_fn(Hl, wq, last_flags & UMX_FLAG_PRECISE_ACT);
    const int threads = groups > 1 && !grouped ? LSTMB2_THREADS : LSTM_THREADS;
    const int Sw = grouped ? groups * (S / kBatchsSpan) : S; // workgroups per chain of the launch's grid
    // the twelve-wave kernel of lstm_batch2.h takes the row sums but leaves the planes to split_planes_kernel
    const bool writes_planes = fuse && lstm_writes_planes && (groups == 1 || grouped);
    if (!writes_planes)
        for (int i = 0; i < 4; ++i)
            a.planes[i] = nullptr;
    a.write_f32 = (last_flags & UMX_FLAG_DEBUG_TAPS) ? 1 : 0;
    sl.lstm_wrote_planes[layer] = writes_planes;
    void *kargs[] = {&a};
    bool persistent = !stepwise && persistent_ok && 8 * S <= lstm_batch_capacity;
    if (persistent)
    {
        a.tag_epoch = next_tag_base() >> 12; // unique per launch (20 bits); the granule area is cleared when it wraps
        const bool clear = tag_epoch == 0;
        a.t_begin = 0;
        a.t_end = T;
        a.census = 1;
        a.abort_at = ((last_flags & UMX_FLAG_DEBUG_LSTM_ABORT) && layer == 1) ? T / 2 : 0;
        UMX_HIP_CHECK(hipMemsetAsync(sl.lsync, 0, sizeof(unsigned) * (clear ? lsync_words : LSTM_SYNC_HEADER_WORDS), st));
        hipError_t e = lstm_gate_launch(device, st, 8 * S * 2, 2 * n_cus,
                                        [&] { return hipLaunchKernel(fn, dim3(8 * Sw), dim3(threads), kargs, lds, st); });
        if (e != hipSuccess)
        {
            (void)hipGetLastError();
            persistent_ok = false;
            persistent = false;
        }
    }
    if (!persistent)
    {
        a.census = 0;
        a.abort_at = 0;
        // One step per launch, h / c carried through the fp32 stream state.  Every workgroup reads the h of its whole chain at
        // the start of a launch and writes its own units at the end, so a launch must not update the copy it reads: the
        // workgroups of a grid do not start together (round 3: one GPU-suite run in nine failed the per-step bitwise test at
        // 20 lanes; rounds 2-3 updated `state` in place).  This layer's entries ping-pong between `state` and `state_alt`
        // (rows = lanes x targets, 4 Hl floats of every 12 Hl: the other layers' entries may be in use by another slot).
        const size_t row = (size_t)4 * Hl * sizeof(float), pitch = 3 * row, off = (size_t)layer * 4 * Hl;
        UMX_HIP_CHECK(hipMemcpy2DAsync(state_alt + off, pitch, state + off, pitch, row, (size_t)4 * B, hipMemcpyDeviceToDevice, st));
        for (int step = 0; step < T; ++step)
        {
            a.t_begin = step;
            a.t_end = step + 1;
            a.state = (step & 1) ? state_alt : state;
            a.state_out = (step & 1) ? state : state_alt;
            UMX_HIP_CHECK(hipLaunchKernel(fn, dim3(2 * nact * Sw), dim3(threads), kargs, lds, st));
        }
        if (T & 1) // the last launch wrote state_alt
            UMX_HIP_CHECK(hipMemcpy2DAsync(state + off, pitch, state_alt + off, pitch, row, (size_t)4 * B, hipMemcpyDeviceToDevice, st));
    }
    if (fuse) // the one row per direction that no later step multiplied with
        hipLaunchKernelGGL(lstm_last_row_sum_kernel, dim3(top, 2 * nact), dim3(64), 0, st, a, nact);
    sl.last_persistent = persistent;
    UMX_HIP_CHECK(hipGetLastError());
    return UMX_OK;
}

// ---------------------------------------------------------------- stages of one segment
void umx_hip_ctx::launch_gemm(Lane &sl, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg)
{
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.M = Tp;
    g.T = T;
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const TargetAct &c = sl.ta[active[i]];
        GemmTarget &t = g.t[i];
        switch (mode)
        {
        case G_FC1:
            t.A = sl.x; t.B = b.fc1_w; t.C = c.cat;
            t.e0 = b.bn1[0]; t.e1 = b.bn1[1]; t.e2 = b.bn1[2]; t.e3 = b.bn1[3];
            t.q0 = b.in_scale; t.q1 = b.in_mean;
            g.N = H; g.K = KX; g.lda = KX; g.ldc = 2 * H;
            break;
        case G_IH:
            t.A = layer == 0 ? c.cat : layer == 1 ? c.la : c.lb;
            t.B = b.ih_w[layer]; t.C = c.P; t.e0 = b.ih_b[layer];
            g.N = 4 * H; g.K = H; g.lda = layer == 0 ? 2 * H : H; g.ldc = 4 * H;
            break;
        case G_FC2:
            t.A = c.cat; t.B = b.fc2_w; t.C = c.a2;
            t.e0 = b.bn2[0]; t.e1 = b.bn2[1]; t.e2 = b.bn2[2]; t.e3 = b.bn2[3];
            g.N = H; g.K = 2 * H; g.lda = 2 * H; g.ldc = H;
            break;
        default:
            t.A = c.a2; t.B = b.fc3_w; t.C = c.mag;
            t.e0 = b.bn3[0]; t.e1 = b.bn3[1]; t.e2 = b.bn3[2]; t.e3 = b.bn3[3];
            t.q0 = b.out_scale; t.q1 = b.out_mean;
            g.N = NOUT_PAD; g.K = H; g.lda = H; g.ldc = 0;
            break;
        }
    }
    // quantised-resident B (config 5): all active targets were loaded the same way
    int bq = BQ_F32;
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const QMat &q = mode == G_FC1 ? b.fc1_q : mode == G_IH ? b.ih_q[layer] : mode == G_FC2 ? b.fc2_q : b.fc3_q;
        GemmTarget &t = g.t[i];
        t.bsplit = mode == G_IH ? 2 * H : 0x7fffffff; // W_ih rows >= 4*Hl belong to the reverse direction's tensor
        t.bs[0] = t.bs[1] = 1.f;
        if (q.q)
        {
            t.Bq = q.q;
            t.bs[0] = q.s[0]; t.bs[1] = q.s[1];
            t.bo[0] = q.o[0]; t.bo[1] = q.o[1];
            bq = q.type;
        }
    }
    if (gemm_bf16x3 && bq == BQ_F32) // weights resident as three bf16 planes
        for (int i = 0; i < nact; ++i)
        {
            const TargetBufs &b = tb[active[i]];
            g.t[i].Bq = mode == G_FC1 ? b.fc1_bx : mode == G_IH ? b.ih_bx[layer] : mode == G_FC2 ? b.fc2_bx : b.fc3_bx;
        }
    const dim3 grid((unsigned)round_up((g.N / GEMM_BN) * (g.M / GEMM_BM), 8), 1, nact), block(256);
#define UMX_LAUNCH(KERNEL, LDS) hipLaunchKernelGGL((KERNEL), grid, block, LDS, st, g)
    switch (mode)
        {
        case G_FC1:
            if (bq == BQ_U8 && !u8_dequant) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC1, BQ_U8X>), BX_LDS_BYTES);
            else if (bq == BQ_U8) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC1, BQ_U8>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_FC1, BQ_F32>), BX_LDS_BYTES);
            break;
        case G_IH:
            if (bq == BQ_U8 && !u8_dequant) UMX_LAUNCH((gemm_bf16x3_kernel<G_IH, BQ_U8X>), BX_LDS_BYTES);
            else if (bq == BQ_U8) UMX_LAUNCH((gemm_bf16x3_kernel<G_IH, BQ_U8>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_IH, BQ_F32>), BX_LDS_BYTES);
            break;
        case G_FC2:
            if (bq == BQ_U16) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC2, BQ_U16>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_FC2, BQ_F32>), BX_LDS_BYTES);
            break;
        default:
            if (bq == BQ_U16) UMX_LAUNCH((gemm_bf16x3_kernel<G_FC3, BQ_U16>), BX_LDS_BYTES);
            else UMX_LAUNCH((gemm_bf16x3_kernel<G_FC3, BQ_F32>), BX_LDS_BYTES);
            break;
        }
#undef UMX_LAUNCH
}

// gemm_planes.h: split one A operand of every active target into bf16 planes + row sums
// ln = the first of nl consecutive track lanes (their buffers are contiguous: see init)
void umx_hip_ctx::launch_split(Lane &ln, int nl, hipStream_t st, int which, const int *active, int nact)
{
    SplitArgs a;
    memset(&a, 0, sizeof a);
    a.T = T;
    a.Tp = Tp;
    const size_t rows_all = (size_t)B * Tp + Mpad; // rows of one output plane
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const TargetAct &c = ln.ta[active[i]];
        switch (which)
        {
        case SP_XS: // x * input_scale + input_mean (inference.cpp:78-83), per target
            a.src[i] = ln.x; a.dst[i] = c.xs_p; a.rowsum[i] = c.rs_xs; a.rowunscale[i] = c.rsc_xs; a.scale[i] = b.in_scale; a.mean[i] = b.in_mean;
            a.cols = KX; a.ld_src = KX; a.ld_dst = KX; a.col0_dst = 0; a.plane = rows_all * KX;
            break;
        case SP_CATL: // fc1 output = left half of the skip concat
            a.src[i] = c.cat; a.dst[i] = c.cat_p; a.rowsum[i] = c.rs_catL;
            a.cols = H; a.ld_src = 2 * H; a.ld_dst = 2 * H; a.col0_dst = 0; a.plane = rows_all * 2 * H;
            break;
        case SP_CATR: // last LSTM layer's output = right half
            a.src[i] = c.cat + H; a.dst[i] = c.cat_p; a.rowsum[i] = lstm_rowsums ? nullptr : c.rs_catR;
            a.cols = H; a.ld_src = 2 * H; a.ld_dst = 2 * H; a.col0_dst = H; a.plane = rows_all * 2 * H;
            break;
        case SP_LA:
            a.src[i] = c.la; a.dst[i] = c.la_p; a.rowsum[i] = lstm_rowsums ? nullptr : c.rs_la;
            a.cols = H; a.ld_src = H; a.ld_dst = H; a.col0_dst = 0; a.plane = rows_all * H;
            break;
        case SP_LB:
            a.src[i] = c.lb; a.dst[i] = c.lb_p; a.rowsum[i] = lstm_rowsums ? nullptr : c.rs_lb;
            a.cols = H; a.ld_src = H; a.ld_dst = H; a.col0_dst = 0; a.plane = rows_all * H;
            break;
        default:
            a.src[i] = c.a2; a.dst[i] = c.a2_p; a.rowsum[i] = c.rs_a2; a.rowunscale[i] = c.rsc_a2;
            a.cols = H; a.ld_src = H; a.ld_dst = H; a.col0_dst = 0; a.plane = rows_all * H;
            break;
        }
    }
    a.rows_valid = nl * Tp;
    const dim3 grid(round_up(nl * Tp, 256) / 4, 1, nact);
    if (which == SP_XS && a.cols <= 3072) // every target's operand comes from the same rows of x: read them once
    {
        hipLaunchKernelGGL(split_planes_shared_kernel<6>, dim3(grid.x), dim3(256), 0, st, a, nact);
        return;
    }
    if (a.cols <= 1024)
        hipLaunchKernelGGL(split_planes_kernel<2>, grid, dim3(256), 0, st, a);
    else if (a.cols <= 3072)
        hipLaunchKernelGGL(split_planes_kernel<6>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(split_planes_kernel<8>, grid, dim3(256), 0, st, a);
}

void umx_hip_ctx::launch_gemm_planes(Lane &ln, int nl, hipStream_t st, int mode, int layer, const int *active, int nact, bool dbg)
{
    GemmPArgs g;
    memset(&g, 0, sizeof g);
    g.M = round_up(nl * Tp, 256); // rows behind the last lane of the launch are zero planes (split_planes_kernel)
    g.lanes = nl;
    g.T = T;
    g.Tp_lane = Tp;
    g.mag_lane = (size_t)2 * T * MAGP;
    g.a_unscale = 1.0f / (float)(1 << GP_SPLIT_FIXED_EXP); // tanh / LSTM outputs: constant scale (split_planes_kernel)
    const size_t rows_all = (size_t)B * Tp + Mpad;
    int nbp = 2;
    for (int i = 0; i < nact; ++i)
    {
        const TargetBufs &b = tb[active[i]];
        const TargetAct &c = ln.ta[active[i]];
        GemmPTarget &t = g.t[i];
        const PMat *pm = nullptr;
        t.bsplit = 0x7fffffff;
        switch (mode)
        {
        case G_FC1:
            pm = &b.fc1_p;
            t.A = c.xs_p; t.C = c.cat; t.rs0 = c.rs_xs; t.rsc = c.rsc_xs;
            t.e0 = b.bn1[0]; t.e1 = b.bn1[1]; t.e2 = b.bn1[2]; t.e3 = b.bn1[3];
            g.N = H; g.K = KX; g.lda = KX; g.ldc = 2 * H; g.a_plane = rows_all * KX;
            break;
        case G_IH:
            pm = &b.ih_p[layer];
            t.A = layer == 0 ? c.cat_p : layer == 1 ? c.la_p : c.lb_p;
            t.rs0 = layer == 0 ? c.rs_catL : layer == 1 ? c.rs_la : c.rs_lb;
            if (layer > 0 && lstm_rowsums) // one row-sum array per direction, from the recurrence itself
                t.rs1 = t.rs0 + rows_all;
            t.C = c.P; t.e0 = b.ih_b[layer];
            t.bsplit = 2 * H; // W_ih rows >= 4*Hl belong to the reverse direction's tensor
            g.N = 4 * H; g.K = H; g.lda = layer == 0 ? 2 * H : H; g.ldc = 4 * H;
            g.a_plane = layer == 0 ? rows_all * 2 * H : rows_all * H;
            break;
        case G_FC2:
            pm = &b.fc2_p;
            t.A = c.cat_p; t.C = c.a2; t.rs0 = c.rs_catL; t.rs1 = c.rs_catR;
            if (lstm_rowsums)
                t.rs2 = c.rs_catR + rows_all;
            t.e0 = b.bn2[0]; t.e1 = b.bn2[1]; t.e2 = b.bn2[2]; t.e3 = b.bn2[3];
            g.N = H; g.K = 2 * H; g.lda = 2 * H; g.ldc = H; g.a_plane = rows_all * 2 * H;
            break;
        default:
            pm = &b.fc3_p;
            t.A = c.a2_p; t.C = c.mag; t.rs0 = c.rs_a2; t.rsc = c.rsc_a2;
            t.e0 = b.bn3[0]; t.e1 = b.bn3[1]; t.e2 = b.bn3[2]; t.e3 = b.bn3[3];
            t.q0 = b.out_scale; t.q1 = b.out_mean;
            g.N = NOUT_PAD; g.K = H; g.lda = H; g.ldc = 0; g.a_plane = rows_all * H;
            break;
        }
        t.B = pm->p;
        t.bs[0] = pm->s[0]; t.bs[1] = pm->s[1];
        t.bo2[0] = pm->o2[0]; t.bo2[1] = pm->o2[1];
        nbp = pm->nbp; // the same for every target (all_q at create)
    }
    // 256 x 256 tiles (half the L2 traffic per flop) when they fill the chip
    const int blocks_big = (g.N / 256) * (g.M / 256) * nact;
    const bool big = g.N % 256 == 0 && blocks_big >= 224;
    const int bm = big ? 256 : 128;
    const dim3 grid((unsigned)round_up((g.N / bm) * (g.M / bm), 8), 1, nact), block(big ? 1024 : 256);
    const size_t lds = big ? gp_lds_bytes(4, 4, nbp) : gp_lds_bytes(2, 2, nbp);
    // 256 x 256 blocks: eight waves of 128 x 64 in ping-pong (gemm_planes_pp.h), or sixteen waves of 64 x 64 in lock step
    // (gemm_planes.h: same bits; UMX_GEMM_PP=0, or a bit per GemmMode).  Measured alone, 32 lanes, ms per launch incl. the split
    // kernel, A/B on one box (round 3): fc1 5.70-5.98 -> 5.38-5.52, W_ih 5.69-6.14 -> 5.29-5.67, fc2 4.77 -> 4.62-4.66, fc3 9.21 -> 8.98-9.15.
    const int gemm_pp = getenv("UMX_GEMM_PP") ? atoi(getenv("UMX_GEMM_PP")) : -1; // read per launch: the tests switch it
    const bool pp = big && (gemm_pp < 0 || ((gemm_pp >> mode) & 1));
#define UMX_GP(MODE)                                                                                                 \
    if (pp && nbp == 1) hipLaunchKernelGGL((gemm_planes_pp_kernel<MODE, 1>), grid, dim3(512), lds, st, g);           \
    else if (pp) hipLaunchKernelGGL((gemm_planes_pp_kernel<MODE, 2>), grid, dim3(512), lds, st, g);                  \
    else if (big && nbp == 1) hipLaunchKernelGGL((gemm_planes_kernel<MODE, 1, 4, 4>), grid, block, lds, st, g);      \
    else if (big) hipLaunchKernelGGL((gemm_planes_kernel<MODE, 2, 4, 4>), grid, block, lds, st, g);                  \
    else if (nbp == 1) hipLaunchKernelGGL((gemm_planes_kernel<MODE, 1, 2, 2>), grid, block, lds, st, g);             \
    else hipLaunchKernelGGL((gemm_planes_kernel<MODE, 2, 2, 2>), grid, block, lds, st, g);
    switch (mode)
    {
    case G_FC1: UMX_GP(G_FC1) break;
    case G_IH: UMX_GP(G_IH) break;
    case G_FC2: UMX_GP(G_FC2) break;
    default: UMX_GP(G_FC3) break;
    }
#undef UMX_GP
}

void umx_hip_ctx::launch_gemm_lanes(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, int mode, int layer,
                                    const int *active, int nact, bool dbg)
{
    if (nact <= 0)
        return;
    for (int l0 = 0; l0 < nb;)
    {
        if (!audio_dev[l0])
        {
            ++l0;
            continue;
        }
        int l1 = l0 + 1;
        while (gemm_planes && l1 < nb && audio_dev[l1])
            ++l1;
        if (gemm_planes)
        {
            // the A operand is split here, right before its consumer (every producer -- STFT, fc1, the LSTM layers,
            // fc2 -- writes fp32)
            const int which = mode == G_FC1 ? SP_XS : mode == G_IH ? (layer == 0 ? SP_CATL : layer == 1 ? SP_LA : SP_LB) : mode == G_FC2 ? SP_CATR : SP_A2;
            const int from_layer = which == SP_LA ? 0 : which == SP_LB ? 1 : which == SP_CATR ? 2 : -1;
            if (from_layer < 0 || !sl.lstm_wrote_planes[from_layer]) // else: written by the recurrence
                launch_split(sl.lane[l0], l1 - l0, st, which, active, nact);
            {
                // the stage's kernel time without its split kernel (umx_hip_stage_kernel_times): stage event ... this event ... next stage event
                const int stg = mode == G_FC1 ? ST_FC1 : mode == G_IH ? ST_IH0 + 2 * layer : mode == G_FC2 ? ST_FC2 : ST_FC3;
                sl.evk_set[stg] = sl.evk[stg] && hipEventRecord(sl.evk[stg], st) == hipSuccess;
            }
            launch_gemm_planes(sl.lane[l0], l1 - l0, st, mode, layer, active, nact, dbg);
        }
        else
            launch_gemm(sl.lane[l0], st, mode, layer, active, nact, dbg);
        l0 = l1;
    }
}

// stft -> |.|, crop/stack -> fc1/bn1/tanh -> input projection of LSTM layer 0; stage by stage over the track lanes
int umx_hip_ctx::stage_front(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, const int *n, const int *active,
                             int nact)
{
    stage_range(ST_STFT);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_STFT], st));
    {
        StftIn in;
        in.lanes = lane_set(nb, audio_dev);
        for (int i = 0; i < in.lanes.count; ++i)
        {
            in.audio[i] = audio_dev[in.lanes.id[i]];
            in.n[i] = n[in.lanes.id[i]];
        }
        Lane &L0 = sl.lane[0];
        UMX_HIP_CHECK(hipMemsetAsync(L0.maxabs, 0, sizeof(unsigned) * B, st)); // per-call scratch of every lane
        hipLaunchKernelGGL(stft_kernel, dim3(T, in.lanes.count), dim3(256), 0, st, in, N, T, window, tw1, tw2, L0.spec, lane_strides().spec, L0.x,
                           (size_t)Tp * KX, L0.maxabs);
    }
    stage_range(ST_FC1);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_FC1], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_FC1, 0, active, nact, false);
    stage_range(ST_IH0);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_IH0], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_IH, 0, active, nact, false);
    return UMX_OK;
}

// fc2/bn2/relu -> fc3/bn3/scale/relu/mask -> Wiener (or mix phase) -> iSTFT -> overlap-add
int umx_hip_ctx::stage_back(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n,
                            unsigned flags, const int *active, int nact)
{
    if (int rc = stage_masks(sl, st, nb, audio_dev, flags, active, nact))
        return rc;
    return stage_finish(sl, st, nb, audio_dev, out, n, flags, true);
}

int umx_hip_ctx::stage_masks(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, unsigned flags, const int *active, int nact)
{
    const bool dbg = flags & UMX_FLAG_DEBUG_TAPS;
    stage_range(ST_FC2);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_FC2], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_FC2, 0, active, nact, dbg);
    stage_range(ST_FC3);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_FC3], st));
    launch_gemm_lanes(sl, st, nb, audio_dev, G_FC3, 0, active, nact, dbg);
    UMX_HIP_CHECK(hipGetLastError());
    return UMX_OK;
}

int umx_hip_ctx::stage_finish(Slot &sl, hipStream_t st, int nb, const float *const *audio_dev, float *const *out, const int *n,
                              unsigned flags, bool zero_skipped)
{
    const bool dbg = flags & UMX_FLAG_DEBUG_TAPS;
    if (zero_skipped)
        for (int ln = 0; ln < nb; ++ln)
            if (audio_dev[ln])
                for (int tg = 0; tg < 4; ++tg) // a skipped target contributes an all-zero magnitude
                    if (flags & UMX_FLAG_SKIP_TARGET(tg))
                        UMX_HIP_CHECK(hipMemsetAsync(sl.lane[ln].ta[tg].mag, 0, sizeof(float) * 2 * T * MAGP, st));
    stage_range(ST_WIENER);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_WIENER], st));
    const int bt = (NBINS + 255) / 256;
    const LaneSet lanes = lane_set(nb, audio_dev);
    const WienerStrides ls = lane_strides();
    Lane &L0 = sl.lane[0]; // the batched kernels take lane 0's pointers and step by ls
    WienerMags wm0;
    for (int s = 0; s < 4; ++s)
        wm0.m[s] = L0.ta[s].mag;
    const int nchunk = (T + WIENER_CHUNK - 1) / WIENER_CHUNK;
    if (flags & UMX_FLAG_NO_WIENER)
    {
        if (!wiener_fused)
            for (int i = 0; i < lanes.count; ++i)
            {
                Lane &L = sl.lane[lanes.id[i]];
                WienerMags wm;
                for (int s = 0; s < 4; ++s)
                    wm.m[s] = L.ta[s].mag;
                const size_t nel = (size_t)2 * T * NBINS;
                hipLaunchKernelGGL(mixphase_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, L.spec, wm, T, L.y);
            }
    }
    else
    {
        // two sources per thread (measured: 4 per thread 0.151, 2: 0.128, 1: 0.131 ms per track)
        hipLaunchKernelGGL(wiener_stats4_kernel<2>, dim3((NBINS + 63) / 64, nchunk * lanes.count, 2), dim3(64), 0, st, L0.spec, wm0, T, L0.maxabs, L0.wpart, lanes, ls);
        hipLaunchKernelGGL(wiener_finish4_kernel, dim3(bt, 4, lanes.count), dim3(256), 0, st, L0.wpart, T, L0.Rc, lanes, ls);
        if (!wiener_fused)
            for (int i = 0; i < lanes.count; ++i)
            {
                Lane &L = sl.lane[lanes.id[i]];
                WienerMags wm;
                for (int s = 0; s < 4; ++s)
                    wm.m[s] = L.ta[s].mag;
                hipLaunchKernelGGL(wiener_apply_kernel, dim3(bt, T), dim3(256), 0, st, L.spec, wm, T, L.maxabs, L.Rc, L.y);
            }
    }
    OlaOut oo;
    oo.lanes = lanes;
    int nmax = 1;
    for (int i = 0; i < lanes.count; ++i)
    {
        const int ln = lanes.id[i];
        for (int s = 0; s < 4; ++s)
            oo.p[i][s] = out[4 * ln + s];
        oo.n[i] = n[ln];
        nmax = std::max(nmax, n[ln]);
    }
    stage_range(ST_ISTFT);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_ISTFT], st));
    if (!wiener_fused)
    {
        for (int i = 0; i < lanes.count; ++i)
        {
            Lane &L = sl.lane[lanes.id[i]];
            hipLaunchKernelGGL(istft_frames_kernel, dim3(T, 4), dim3(256), 0, st, L.y, T, window, nw, tw1, tw2, L.frames);
        }
    }
    else
    {
        // gains + filter + inverse STFT frame + overlap-add in one pass (wiener_istft.h); y reaches HBM only for the debug
        // tap, the frames only at the seams between the runs of frames the workgroups take
        float2 *ydbg = dbg ? L0.y : nullptr;
        if (sl.out_free_valid) // the stems this slot wrote two calls ago are still being downloaded from the same buffers
            UMX_HIP_CHECK(hipStreamWaitEvent(st, sl.out_free, 0));
        // runs: a few rounds of workgroups over the chip (one workgroup per CU), at least three frames each
        const int runs = std::max(1, std::min(T / 8, (4 * n_cus + lanes.count - 1) / lanes.count));
        const int run_len = std::max(3, (T + runs - 1) / runs), nruns = (T + run_len - 1) / run_len;
        // one 1024-thread workgroup per run, all four sources (two / one source per workgroup, i.e. more workgroups per CU that
        // each repeat the source-independent part, measured 1.7x / 2.7x slower in round 2)
        if (flags & UMX_FLAG_NO_WIENER)
            hipLaunchKernelGGL((wiener_istft_kernel<false>), dim3(nruns, 1, lanes.count), dim3(1024), (size_t)4 * FFT_LDS_ELEMS * sizeof(float2), st, L0.spec, wm0,
                               T, L0.maxabs, L0.Rc, window, nw, tw1, tw2, L0.frames, ydbg, ls, run_len, oo);
        else
            hipLaunchKernelGGL((wiener_istft_kernel<true>), dim3(nruns, 1, lanes.count), dim3(1024), (size_t)4 * FFT_LDS_ELEMS * sizeof(float2), st, L0.spec, wm0,
                               T, L0.maxabs, L0.Rc, window, nw, tw1, tw2, L0.frames, ydbg, ls, run_len, oo);
        stage_range(ST_OLA);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_OLA], st));
        hipLaunchKernelGGL(wiener_ola_edges_kernel, dim3(3 * HOP / 256, nruns * 4, lanes.count), dim3(256), 0, st, L0.frames, ls.frames, T, run_len, oo);
    }
    if (!wiener_fused)
    {
        stage_range(ST_OLA);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_OLA], st));
        if (sl.out_free_valid) // the stems this slot wrote two calls ago are still being downloaded from the same buffers
            UMX_HIP_CHECK(hipStreamWaitEvent(st, sl.out_free, 0));
        hipLaunchKernelGGL(istft_ola_kernel, dim3((nmax + 255) / 256, 4, lanes.count), dim3(256), 0, st, L0.frames, ls.frames, T, oo);
    }
    stage_range(-1);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_COUNT], st));
    UMX_HIP_CHECK(hipGetLastError());
    sl.have_times = true;
    return UMX_OK;
}

// ---------------------------------------------------------------- one segment (of every track lane)
int umx_hip_ctx::infer_device(const float *audio_dev, int n, float *const out[4], unsigned flags)
{
    if (!audio_dev)
    {
        set_error("infer_segment: need 1 <= n <= segment_samples and non-null audio");
        return UMX_ERR_ARG;
    }
    return infer_batch(1, &audio_dev, &n, out, flags);
}

int umx_hip_ctx::infer_batch(int nb, const float *const *audio_dev, const int *n, float *const *out, unsigned flags)
{
    if (nb < 1 || nb > B || !audio_dev || !n || !out)
    {
        set_error("infer: need 1 <= n_tracks <= the context's track count and non-null argument arrays");
        return UMX_ERR_ARG;
    }
    unsigned long long lane_mask = 0;
    for (int ln = 0; ln < nb; ++ln)
    {
        if (!audio_dev[ln]) // idle lane: its stream state stays as it is
            continue;
        if (n[ln] < 1 || n[ln] > N)
        {
            set_error("infer_segment: need 1 <= n <= segment_samples and non-null audio");
            return UMX_ERR_ARG;
        }
        for (int s = 0; s < 4; ++s)
            if (!out[4 * ln + s])
            {
                set_error("infer_segment: null output pointer");
                return UMX_ERR_ARG;
            }
        lane_mask |= 1ull << ln;
    }
    if (!lane_mask)
    {
        set_error("infer: no active track lane");
        return UMX_ERR_ARG;
    }
    if (ph_next != -1)
    {
        set_error("infer_segment: a phased segment is open (umx_hip_segment_end first)");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    // Consecutive calls alternate between two slots/streams; a slot is reused two calls later (stream order
    // protects its buffers).  Everything that touches the streaming LSTM state is ordered by events: R_l of
    // this segment waits for R_l of the previous one.
    const int si = next_slot();
    Slot &sl = slot[si];
    Slot &prev = slot[(si + nslots - 1) % nslots];
    hipStream_t st = sl.stream;
    int active[4], nact;
    active_list(flags, active, nact);
    last_flags = flags;
    const size_t call_idx = pending_lost ? (size_t)kBackupCalls : pending.size();
    if (call_idx >= (size_t)kBackupCalls)
    {
        pending_lost = true; // no state backup left for this call: a timeout before the next sync cannot be repaired
        pending.clear();
    }
    else
    {
        PendingCall pc;
        memset(&pc, 0, sizeof pc);
        pc.nb = nb;
        pc.flags = flags;
        for (int ln = 0; ln < nb; ++ln)
        {
            pc.audio[ln] = audio_dev[ln];
            pc.n[ln] = n[ln];
            for (int s2 = 0; s2 < 4; ++s2)
                pc.out[4 * ln + s2] = out[4 * ln + s2];
        }
        pending.push_back(pc);
    }
    // Track-batched contexts: the kernels of consecutive calls run one after the other.  Their workgroups take whole CUs (plane
    // GEMM, batched LSTM, fused Wiener kernel), so kernels of two calls side by side only wait for each other's CUs: since the
    // streaming kernels cover all lanes in one launch the serial step is the faster one (32 lanes: 75.4 against 76.3 ms).  The two
    // slots remain for the buffers: uploads and downloads of neighbouring calls still overlap these kernels.
    if (lstm_batched && prev.used && &prev != &sl)
        UMX_HIP_CHECK(hipStreamWaitEvent(st, prev.ev[ST_COUNT], 0));
    if (int rc = stage_front(sl, st, nb, audio_dev, n, active, nact))
        return rc;
    // two LSTM grids at once only where both fit (the single-track kernel); otherwise wait for the previous
    // segment's last layer
    const bool two_grids = !lstm_batched && 2 * 8 * S <= lstm_capacity;
    for (int layer = 0; layer < 3; ++layer)
    {
        if (layer > 0)
        {
            stage_range(ST_IH0 + 2 * layer);
            UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_IH0 + 2 * layer], st));
            launch_gemm_lanes(sl, st, nb, audio_dev, G_IH, layer, active, nact, false);
        }
        if (prev.used) // the previous segment's layer `layer` must have left its final h/c (F3)
            UMX_HIP_CHECK(hipStreamWaitEvent(st, prev.rec_done[two_grids ? layer : 2], 0));
        stage_range(ST_LSTM0 + 2 * layer);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_LSTM0 + 2 * layer], st));
        if (call_idx < (size_t)kBackupCalls) // the state this layer starts from (the previous segment's layer has finished)
            UMX_HIP_CHECK(hipMemcpyAsync(backup + (call_idx * 3 + layer) * state_floats() * B, state, sizeof(float) * state_floats() * B,
                                         hipMemcpyDeviceToDevice, st));
        if (nact > 0)
            if (int rc = run_lstm_layer(sl, layer, active, nact, flags & UMX_FLAG_LSTM_STEPWISE, lane_mask))
                return rc;
        UMX_HIP_CHECK(hipEventRecord(sl.rec_done[layer], st));
    }
    if (int rc = stage_back(sl, st, nb, audio_dev, out, n, flags, active, nact))
        return rc;
    sl.used = true;
    cur = si;
    ++nseg;
    return UMX_OK;
}

// ---------------------------------------------------------------- whole track
// shift_inference (umx.cpp:99-150) around split_inference (umx.cpp:152-295) with the track resident in HBM:
// one upload, the segments queued back to back through the two pipeline slots (so consecutive segments
// overlap exactly as in bench.py), the weighted overlap-add and the final normalisation on the device, one
// download.  shift_offset < 0: no shift buffer (plain split_inference).
int umx_hip_ctx::track(const float *audio_host, int length, int shift_offset, float *const out_host[4], unsigned flags,
                       void (*progress)(float, void *), void *progress_user)
{
    if (!out_host)
    {
        set_error("track: need audio, outputs, length >= 1 and shift offset < 22050");
        return UMX_ERR_ARG;
    }
    return tracks(1, &audio_host, &length, &shift_offset, out_host, flags, progress, progress_user);
}

// shift_inference (umx.cpp:99-150) around split_inference (umx.cpp:152-295) for `nt` tracks at once, one per track
// lane: the tracks stay in HBM, call s runs segment s of every track that still has one (a finished track's lane sits
// idle), the weighted overlap-add and the normalisation run per lane on the device, finished regions are downloaded
// while later segments run.  nt == 1 is umx_hip_split_inference / umx_hip_shift_inference.
int umx_hip_ctx::tracks(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host,
                        unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    // A persistent-kernel timeout inside a track cannot be repaired segment by segment (the overlap-add has consumed
    // the stems): everything is run again, once, with the per-step driver the timeout switches the context to.
    no_recovery = true;
    int rc = tracks_once(nt, audio_host, length, shift_offset, out_host, flags, progress, progress_user);
    if (rc == UMX_ERR_TIMEOUT)
        rc = tracks_once(nt, audio_host, length, shift_offset, out_host, flags, progress, progress_user);
    no_recovery = false;
    pending.clear();
    pending_lost = false;
    return rc;
}

int umx_hip_ctx::tracks_once(int nt, const float *const *audio_host, const int *length, const int *shift_offset, float *const *out_host,
                             unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    if (nt < 1 || nt > B || !audio_host || !length || !shift_offset || !out_host)
    {
        set_error("tracks: need 1 <= n_tracks <= the context's track count and non-null argument arrays");
        return UMX_ERR_ARG;
    }
    for (int ln = 0; ln < nt; ++ln)
        if (!audio_host[ln] || length[ln] < 1 || shift_offset[ln] >= UMX_MAX_SHIFT || !out_host[4 * ln] || !out_host[4 * ln + 1] ||
            !out_host[4 * ln + 2] || !out_host[4 * ln + 3])
        {
            set_error("track: need audio, outputs, length >= 1 and shift offset < 22050");
            return UMX_ERR_ARG;
        }
    if (ph_next != -1)
    {
        set_error("track: a phased segment is open (umx_hip_segment_end first)");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    if (int rc = sync_all())
        return rc;
    int lead[LSTMB_MAX_TRACKS], L2[LSTMB_MAX_TRACKS], L2max = 0;
    for (int ln = 0; ln < nt; ++ln)
    {
        lead[ln] = shift_offset[ln] < 0 ? 0 : shift_offset[ln];
        // umx.cpp:120-122: length + max_shift - offset -- which the reference overruns for offset > max_shift / 2 (its
        // block write is [offset, offset + length)); the same size where the reference is defined, large enough elsewhere
        const long long l2 = shift_offset[ln] < 0 ? (long long)length[ln]
                                                  : (long long)length[ln] + std::max(UMX_MAX_SHIFT - shift_offset[ln], shift_offset[ln]);
        if (l2 > 0x7fffffff / 2)
        {
            set_error("track: too long");
            return UMX_ERR_ARG;
        }
        L2[ln] = (int)l2;
        L2max = std::max(L2max, L2[ln]);
    }
    if (trk.size() < (size_t)nt)
        trk.resize(nt);
    for (int ln = 0; ln < nt; ++ln)
    {
        TrackBufs &tb_ = trk[ln];
        if ((size_t)L2[ln] > tb_.cap) // grow-only track buffers
        {
            const size_t cap = (size_t)L2[ln] + (size_t)L2[ln] / 8;
            for (float **p : {&tb_.in, &tb_.out[0], &tb_.out[1], &tb_.out[2], &tb_.out[3], &tb_.sumw})
                if (*p)
                {
                    allocs.erase(std::find(allocs.begin(), allocs.end(), (void *)*p));
                    (void)hipFree(*p);
                    *p = nullptr;
                }
            tb_.cap = 0;
            if (int rc = dalloc(&tb_.in, 2 * cap, false))
                return rc;
            for (int t = 0; t < 4; ++t)
                if (int rc = dalloc(&tb_.out[t], 2 * cap, false))
                    return rc;
            if (int rc = dalloc(&tb_.sumw, cap, false))
                return rc;
            tb_.cap = cap;
        }
        if (!tb_.seg[0][0])
            for (int s = 0; s < nslots; ++s)
                for (int t = 0; t < 4; ++t)
                    if (int rc = dalloc(&tb_.seg[s][t], (size_t)2 * N, false))
                        return rc;
    }
    if (!trk_acc_ev[0])
        for (int s = 0; s < nslots; ++s)
            UMX_HIP_CHECK(hipEventCreateWithFlags(&trk_acc_ev[s], hipEventDisableTiming));
    // umx.cpp:167-171: a fresh, zeroed lstm_data per track; umx.cpp:186-195: zeroed accumulators (and F4)
    UMX_HIP_CHECK(hipMemset(state, 0, sizeof(float) * state_floats() * nt));
    clear_used();
    for (int ln = 0; ln < nt; ++ln)
    {
        TrackBufs &tb_ = trk[ln];
        UMX_HIP_CHECK(hipMemset(tb_.in, 0, sizeof(float) * 2 * (size_t)L2[ln]));
        for (int t = 0; t < 4; ++t)
            UMX_HIP_CHECK(hipMemset(tb_.out[t], 0, sizeof(float) * 2 * (size_t)L2[ln]));
        UMX_HIP_CHECK(hipMemset(tb_.sumw, 0, sizeof(float) * (size_t)L2[ln]));
        UMX_HIP_CHECK(hipMemcpy(tb_.in + 2 * (size_t)lead[ln], audio_host[ln], sizeof(float) * 2 * (size_t)length[ln], hipMemcpyHostToDevice));
    }
    UMX_HIP_CHECK(hipDeviceSynchronize());

    const int stride = (int)((1 - 0.25f) * N); // umx.cpp:181, inference.hpp:15
    const float total_reps = std::ceil((float)L2max / (float)stride); // umx.cpp:208 (of the longest track)
    float done = 0.f;
    // A sample is final once the segment that starts at or before it and the one before that have been blended
    // in: region [offset_i, offset_{i+1}) right after segment i.  It is normalised there and then, and the host
    // downloads it while the GPU is already busy with the following segments.
    struct Region
    {
        int lane, start, count;
        hipEvent_t ready;
    };
    std::vector<Region> regions;
    auto cleanup = [&]() {
        for (Region &r : regions)
            (void)hipEventDestroy(r.ready);
    };
    int last_slot = -1, iseg = 0;
    for (long long off = 0; off < L2max; off += stride, ++iseg)
    {
        const int offset = (int)off;
        const int si = next_slot();
        const float *ain[LSTMB_MAX_TRACKS] = {};
        int nn[LSTMB_MAX_TRACKS] = {};
        float *outs[4 * LSTMB_MAX_TRACKS] = {};
        for (int ln = 0; ln < nt; ++ln)
            if (offset < L2[ln]) // this track still has a segment here (umx.cpp:214-217)
            {
                ain[ln] = trk[ln].in + 2 * (size_t)offset;
                nn[ln] = std::min(N, L2[ln] - offset);
                for (int t = 0; t < 4; ++t)
                    outs[4 * ln + t] = trk[ln].seg[si][t];
            }
        const int rc = infer_batch(nt, ain, nn, outs, flags);
        if (rc)
        {
            cleanup();
            return rc;
        }
        hipStream_t st = slot[si].stream;
        if (last_slot >= 0) // accumulate in segment order (two segments overlap by a quarter)
            (void)hipStreamWaitEvent(st, trk_acc_ev[last_slot], 0);
        for (int ln = 0; ln < nt; ++ln)
        {
            if (!ain[ln])
                continue;
            Stems4 tk, seg;
            for (int t = 0; t < 4; ++t)
            {
                tk.p[t] = reinterpret_cast<float2 *>(trk[ln].out[t]);
                seg.p[t] = reinterpret_cast<float2 *>(trk[ln].seg[si][t]);
            }
            hipLaunchKernelGGL(track_accumulate_kernel, dim3((nn[ln] + 255) / 256, 4), dim3(256), 0, st, tk, trk[ln].sumw, seg, offset, nn[ln], N);
            Region rg;
            rg.lane = ln;
            rg.start = offset;
            rg.count = (int)std::min<long long>(off + stride, L2[ln]) - offset;
            hipLaunchKernelGGL(track_normalise_kernel, dim3((rg.count + 255) / 256, 4), dim3(256), 0, st, tk, trk[ln].sumw, rg.start, rg.count);
            if (hipEventCreateWithFlags(&rg.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(rg.ready, st) != hipSuccess)
            {
                cleanup();
                set_error("track: event creation failed");
                return UMX_ERR_HIP;
            }
            regions.push_back(rg);
        }
        (void)hipEventRecord(trk_acc_ev[si], st);
        last_slot = si;
        done += 1.0f / total_reps; // umx.cpp:229 (queued, not finished: the device runs behind the host here)
        if (progress)
            progress(done, progress_user);
    }
    hipError_t cerr = hipSuccess;
    for (const Region &rg : regions) // umx.cpp:136-147: drop the shift
    {
        const int ln = rg.lane;
        const long long lo = std::max<long long>(rg.start, lead[ln]),
                        hi = std::min<long long>((long long)rg.start + rg.count, (long long)lead[ln] + length[ln]);
        if (hi <= lo)
            continue;
        if (cerr == hipSuccess)
            cerr = hipEventSynchronize(rg.ready);
        for (int t = 0; t < 4 && cerr == hipSuccess; ++t)
            cerr = hipMemcpy(out_host[4 * ln + t] + 2 * (size_t)(lo - lead[ln]), trk[ln].out[t] + 2 * (size_t)lo,
                             sizeof(float) * 2 * (size_t)(hi - lo), hipMemcpyDeviceToHost);
    }
    cleanup();
    if (cerr != hipSuccess)
    {
        set_error(hipGetErrorString(cerr));
        return UMX_ERR_HIP;
    }
    UMX_HIP_CHECK(hipGetLastError());
    if (int rc = umx_hip_sync(this)) // surfaces a persistent-kernel timeout
        return rc;
    return UMX_OK;
}

// ---------------------------------------------------------------- one segment, phase by phase
// The same launches as infer_device on slot 0, cut where another GPU's LSTM state has to come in: the
// caller sets layer l's incoming (h, c) (umx_hip_stream_set_layer) before phase_layer(l) and reads the
// outgoing one after it.  Used by the exact state-carry pipeline over several GPUs (multigpu.py).
int umx_hip_ctx::phase_begin(const float *audio_host, int n, unsigned flags)
{
    if (!audio_host || n < 1 || n > N)
    {
        set_error("segment_begin: need 1 <= n <= segment_samples and non-null audio");
        return UMX_ERR_ARG;
    }
    if (ph_next != -1)
    {
        set_error("segment_begin: a phased segment is already open");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    if (int rc = sync_all())
        return rc;
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(flags, active, nact);
    last_flags = flags;
    UMX_HIP_CHECK(hipMemcpyAsync(audio_in, audio_host, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, sl.stream));
    const float *ain = audio_in;
    if (int rc = stage_front(sl, sl.stream, 1, &ain, &n, active, nact))
        return rc;
    ph_next = 0;
    ph_n = n;
    ph_flags = flags;
    ph_audio = nullptr;
    return UMX_OK;
}

// the same without host transfers and without waiting (multi-GPU driver: everything stays on slot 0's stream)
int umx_hip_ctx::phase_begin_device(const float *audio_dev, int n, unsigned flags)
{
    if (!audio_dev || n < 1 || n > N)
    {
        set_error("segment_begin_device: need 1 <= n <= segment_samples and non-null audio");
        return UMX_ERR_ARG;
    }
    if (ph_next != -1)
    {
        set_error("segment_begin: a phased segment is already open");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(flags, active, nact);
    last_flags = flags;
    if (int rc = stage_front(sl, sl.stream, 1, &audio_dev, &n, active, nact))
        return rc;
    ph_next = 0;
    ph_n = n;
    ph_flags = flags;
    ph_audio = audio_dev;
    return UMX_OK;
}

int umx_hip_ctx::phase_end_device(float *const out_dev_[4])
{
    if (ph_next != 3 || !out_dev_)
    {
        set_error("segment_end: all three LSTM layers must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(ph_flags, active, nact);
    ph_next = -1;
    const float *ain = ph_audio ? ph_audio : audio_in;
    if (int rc = stage_back(sl, sl.stream, 1, &ain, out_dev_, &ph_n, ph_flags, active, nact))
        return rc;
    cur = 0;
    clear_used();
    return UMX_OK;
}

// fc2 + fc3 of the active targets: their magnitudes are in HBM afterwards (umx_hip_target_mag_device)
int umx_hip_ctx::phase_masks()
{
    if (ph_next != 3)
    {
        set_error("segment_masks: all three LSTM layers must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(ph_flags, active, nact);
    const float *ain = ph_audio ? ph_audio : audio_in;
    if (int rc = stage_masks(sl, sl.stream, 1, &ain, ph_flags, active, nact))
        return rc;
    ph_next = 4;
    return UMX_OK;
}

// Wiener + inverse STFT from the magnitudes of all four targets, wherever they came from
int umx_hip_ctx::phase_finish_device(float *const out_dev_[4])
{
    if (ph_next != 4 || !out_dev_)
    {
        set_error("segment_finish: umx_hip_segment_masks_device must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    ph_next = -1;
    const float *ain = ph_audio ? ph_audio : audio_in;
    if (int rc = stage_finish(sl, sl.stream, 1, &ain, out_dev_, &ph_n, ph_flags, false))
        return rc;
    cur = 0;
    clear_used();
    return UMX_OK;
}

int umx_hip_ctx::phase_layer(int layer)
{
    if (ph_next < 0 || ph_next > 2 || layer != ph_next)
    {
        set_error("segment_lstm_layer: layers run in order 0, 1, 2 after segment_begin");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    hipStream_t st = sl.stream;
    int active[4], nact;
    active_list(ph_flags, active, nact);
    if (layer > 0)
    {
        stage_range(ST_IH0 + 2 * layer);
        UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_IH0 + 2 * layer], st));
        {
            const float *ain = ph_audio ? ph_audio : audio_in;
            launch_gemm_lanes(sl, st, 1, &ain, G_IH, layer, active, nact, false);
        }
    }
    stage_range(ST_LSTM0 + 2 * layer);
    UMX_HIP_CHECK(hipEventRecord(sl.ev[ST_LSTM0 + 2 * layer], st));
    if (nact > 0)
        if (int rc = run_lstm_layer(sl, layer, active, nact, ph_flags & UMX_FLAG_LSTM_STEPWISE, 1u))
            return rc;
    UMX_HIP_CHECK(hipEventRecord(sl.rec_done[layer], st));
    ph_next = layer + 1;
    return UMX_OK;
}

int umx_hip_ctx::phase_end(float *const out_host[4])
{
    if (ph_next != 3 || !out_host)
    {
        set_error("segment_end: all three LSTM layers must have run");
        return UMX_ERR_ARG;
    }
    UMX_HIP_CHECK(hipSetDevice(device));
    Slot &sl = slot[0];
    int active[4], nact;
    active_list(ph_flags, active, nact);
    ph_next = -1;
    const float *ain = audio_in;
    if (int rc = stage_back(sl, sl.stream, 1, &ain, out_dev, &ph_n, ph_flags, active, nact))
        return rc;
    cur = 0;
    if (int rc = umx_hip_sync(this))
        return rc;
    for (int s = 0; s < 4; ++s)
        UMX_HIP_CHECK(hipMemcpy(out_host[s], out_dev[s], sizeof(float) * 2 * (size_t)ph_n, hipMemcpyDeviceToHost));
    clear_used(); // drained: nothing for the next segment to wait for
    return UMX_OK;
}

// ---------------------------------------------------------------- debug taps of what is no longer materialised
// |X| (the STFT kernel keeps only the cropped part the network reads), the target magnitude mask x |X| (formed inside the
// Wiener kernels) and the mask in the reference's (T, 4098) shape: computed on demand, with the SAME device functions the
// hot kernels use (mix_magnitude, common.h), so a tap holds the bits the pipeline works with.
__global__ __launch_bounds__(256) void tap_mix_mag_kernel(const float2 *__restrict__ spec, size_t n, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = mix_magnitude(spec[i]);
}
__global__ __launch_bounds__(256) void tap_target_mag_kernel(const float2 *__restrict__ spec, const float *__restrict__ mask, size_t n,
                                                             float *__restrict__ out) // [2][T][2049]
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = mask[(i / NBINS) * MAGP + i % NBINS] * mix_magnitude(spec[i]); // inference.cpp:175-183
}
__global__ __launch_bounds__(256) void tap_mask_kernel(const float *__restrict__ mask, int T, float *__restrict__ out) // [T][4098]
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * NOUT)
        return;
    const int t = (int)(i / NOUT), k = (int)(i % NOUT), c = k >= NBINS ? 1 : 0;
    out[i] = mask[mask_index(c, T, t, k - c * NBINS)];
}

// ---------------------------------------------------------------- debugging: LDS isolation guard
// A victim workgroup fills 36 KB of LDS with a pattern and keeps verifying it for a while; run beside other
// kernels it shows whether anything else writes into its LDS allocation.
__global__ __launch_bounds__(256) void lds_guard_kernel(unsigned *errs, int rounds)
{
    __shared__ unsigned g[9216]; // 36 KB
    const unsigned salt = blockIdx.x * 2654435761u;
    for (int i = threadIdx.x; i < 9216; i += 256)
        g[i] = salt ^ (unsigned)i;
    __syncthreads();
    for (int r = 0; r < rounds; ++r)
    {
        unsigned bad = 0;
        for (int i = threadIdx.x; i < 9216; i += 256)
            bad += (g[i] != (salt ^ (unsigned)i));
        if (bad)
        {
            atomicAdd(errs, bad);
            atomicAdd(errs + 1, 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 9216; i += 256)
            g[i] = salt ^ (unsigned)i;
        __syncthreads();
        __builtin_amdgcn_s_sleep(32);
    }
}

// After a persistent-kernel timeout (everything has drained): put the stream state back to what it was before the
// first call queued since the last sync, layer by layer, and run those calls again with the per-step driver.
int umx_hip_ctx::recover()
{
    std::vector<PendingCall> calls;
    calls.swap(pending);
    const size_t per = state_floats();
    for (int l = 0; l < 3; ++l) // layer l of every (lane, target): 4 * Hl floats every 12 * Hl
        UMX_HIP_CHECK(hipMemcpy2D(state + (size_t)l * 4 * Hl, sizeof(float) * 12 * Hl, backup + (size_t)l * per * B + (size_t)l * 4 * Hl,
                                  sizeof(float) * 12 * Hl, sizeof(float) * 4 * Hl, (size_t)B * 4, hipMemcpyDeviceToDevice));
    // hipMemset / device-to-device hipMemcpy on the null stream return before the device has done them, and the slots'
    // streams are non-blocking (they do not order against the null stream): without this wait the replayed kernels could
    // read the state before it is restored
    UMX_HIP_CHECK(hipStreamSynchronize(nullptr));
    clear_used();
    recovering = true;
    int rc = UMX_OK;
    for (const PendingCall &pc : calls)
    {
        // host-pointer form: the staging buffer this call read has since been overwritten by the call two later (two
        // staging buffers, up to kBackupCalls calls queued) -- upload the caller's audio again, on the stream the replay of
        // this call is about to be queued on (the earlier user of the buffer ran on the same stream or has been waited for)
        for (int ln = 0; ln < pc.nb; ++ln)
            if (pc.host_audio[ln] && pc.audio[ln])
                UMX_HIP_CHECK(hipMemcpyAsync(const_cast<float *>(pc.audio[ln]), pc.host_audio[ln], sizeof(float) * 2 * (size_t)pc.n[ln],
                                             hipMemcpyHostToDevice, slot[next_slot()].stream));
        if ((rc = infer_batch(pc.nb, pc.audio, pc.n, pc.out, (pc.flags | UMX_FLAG_LSTM_STEPWISE) & ~UMX_FLAG_DEBUG_LSTM_ABORT)) != UMX_OK)
            break;
        for (int k = 0; k < 4 * pc.nb; ++k) // the host-pointer forms had copied the failed run's stems out
            if (pc.host_out[k] && pc.audio[k / 4])
                UMX_HIP_CHECK(hipMemcpyAsync(pc.host_out[k], pc.out[k], sizeof(float) * 2 * (size_t)pc.n[k / 4], hipMemcpyDeviceToHost,
                                             slot[cur].stream));
    }
    recovering = false;
    if (rc == UMX_OK)
        rc = sync_all();
    pending.clear();
    pending_lost = false;
    return rc;
}

// ---------------------------------------------------------------- C-ABI
extern "C"
{

int umx_hip_create(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                   const umx_tensor_view *tensors, int n_tensors)
{
    unsigned cf = 0;
    if (const char *e = getenv("UMX_WEIGHTS_RESIDENT")) // lets umx-cli switch without an API change
        if (std::string(e) == "expanded" || std::string(e) == "f32")
            cf |= UMX_CREATE_DEQUANTISE_AT_LOAD;
    if (const char *e = getenv("UMX_GEMM"))
    {
        if (std::string(e) == "f32")
            cf |= UMX_CREATE_GEMM_F32;
        if (std::string(e) == "bf16x3")
            cf |= UMX_CREATE_GEMM_STAGED;
        if (std::string(e) == "planes")
            cf |= UMX_CREATE_GEMM_PLANES;
    }
    if (const char *e = getenv("UMX_LSTM"))
        if (std::string(e) == "batched")
            cf |= UMX_CREATE_LSTM_BATCHED;
    if (const char *e = getenv("UMX_U8"))
        if (std::string(e) == "dequant")
            cf |= UMX_CREATE_U8_DEQUANT;
    return umx_hip_create_ex(out, device, hidden_size, segment_samples, tensors, n_tensors, cf);
}

size_t umx_hip_weight_bytes(const umx_hip_ctx *ctx) { return ctx ? ctx->weight_bytes : 0; }

int umx_hip_create_ex(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                      const umx_tensor_view *tensors, int n_tensors, unsigned create_flags)
{
    return umx_hip_create_tracks(out, device, hidden_size, segment_samples, tensors, n_tensors, create_flags, 1);
}

int umx_hip_create_tracks(umx_hip_ctx **out, int device, int hidden_size, int segment_samples,
                          const umx_tensor_view *tensors, int n_tensors, unsigned create_flags, int n_tracks)
{
    if (!out || !tensors)
    {
        g_create_error = "umx_hip_create: null argument";
        return UMX_ERR_ARG;
    }
    *out = nullptr;
    umx_hip_ctx *c = new umx_hip_ctx;
    int rc = c->init(device, hidden_size, segment_samples, tensors, n_tensors, create_flags, n_tracks);
    if (rc != UMX_OK)
    {
        g_create_error = c->err;
        umx_hip_destroy(c);
        return rc;
    }
    *out = c;
    return UMX_OK;
}

int umx_hip_n_tracks(const umx_hip_ctx *ctx) { return ctx ? ctx->B : 0; }
int umx_hip_pipeline_depth(const umx_hip_ctx *ctx) { return ctx ? ctx->nslots : 0; }
int umx_hip_lstm_is_batched(const umx_hip_ctx *ctx) { return ctx && ctx->lstm_batched ? 1 : 0; }

unsigned umx_hip_debug_f16_bits(float x) { return f16_rne_bits(x); }

void umx_hip_destroy(umx_hip_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (void *p : ctx->allocs)
        (void)hipFree(p);
    for (hipEvent_t e : ctx->trk_acc_ev)
        if (e)
            (void)hipEventDestroy(e);
    if (ctx->order_ev)
        (void)hipEventDestroy(ctx->order_ev);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    for (int si = 0; si < umx_hip_ctx::kMaxSlots; ++si)
        for (hipEvent_t e : {ctx->slot[si].k_done, ctx->slot[si].out_free})
            if (e)
                (void)hipEventDestroy(e);
    for (int si = 0; si < umx_hip_ctx::kMaxSlots; ++si)
    {
        Slot &sl = ctx->slot[si];
        for (int i = 0; i <= ST_COUNT; ++i)
        {
            if (sl.ev[i])
                (void)hipEventDestroy(sl.ev[i]);
            if (i < ST_COUNT && sl.evk[i])
                (void)hipEventDestroy(sl.evk[i]);
        }
        for (int l = 0; l < 3; ++l)
            if (sl.rec_done[l])
                (void)hipEventDestroy(sl.rec_done[l]);
        if (sl.stream)
            (void)hipStreamDestroy(sl.stream);
    }
    delete ctx;
}

const char *umx_hip_last_error(const umx_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

size_t umx_hip_stream_floats(const umx_hip_ctx *ctx) { return ctx ? ctx->state_floats() : 0; }

// track < 0: every lane
int umx_hip_track_stream_reset(umx_hip_ctx *ctx, int track)
{
    if (!ctx || track >= ctx->B)
        return UMX_ERR_ARG;
    // through umx_hip_sync, not a bare stream wait: a timed-out launch among the calls queued so far is noticed (and
    // repaired by replaying them) BEFORE the state is changed, and the replay log starts afresh behind the change
    if (int rc = umx_hip_sync(ctx))
        return rc;
    const size_t per = ctx->state_floats();
    hipError_t e = track < 0 ? hipMemset(ctx->state, 0, sizeof(float) * per * ctx->B)
                             : hipMemset(ctx->state + per * track, 0, sizeof(float) * per);
    // hipMemset returns before the device has done it, and the slots' non-blocking streams do not order against the null
    // stream: the next segment's kernels must not meet the old state (found in round 3: a rare first-run mismatch)
    if (e == hipSuccess)
        e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    ctx->clear_used(); // nothing in flight: no cross-segment dependency to wait for
    return UMX_OK;
}
int umx_hip_stream_reset(umx_hip_ctx *ctx) { return umx_hip_track_stream_reset(ctx, 0); }

int umx_hip_track_stream_get(umx_hip_ctx *ctx, int track, float *host_dst)
{
    if (!ctx || !host_dst || track < 0 || track >= ctx->B)
        return UMX_ERR_ARG;
    if (int rc = umx_hip_sync(ctx))
        return rc;
    const size_t per = ctx->state_floats();
    hipError_t e = hipMemcpy(host_dst, ctx->state + per * track, sizeof(float) * per, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}
int umx_hip_stream_get(umx_hip_ctx *ctx, float *host_dst) { return umx_hip_track_stream_get(ctx, 0, host_dst); }

int umx_hip_track_stream_set(umx_hip_ctx *ctx, int track, const float *host_src)
{
    if (!ctx || !host_src || track < 0 || track >= ctx->B)
        return UMX_ERR_ARG;
    if (int rc = umx_hip_sync(ctx))
        return rc;
    const size_t per = ctx->state_floats();
    hipError_t e = hipMemcpy(ctx->state + per * track, host_src, sizeof(float) * per, hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    ctx->clear_used();
    return UMX_OK;
}
int umx_hip_stream_set(umx_hip_ctx *ctx, const float *host_src) { return umx_hip_track_stream_set(ctx, 0, host_src); }

size_t umx_hip_stream_layer_floats(const umx_hip_ctx *ctx) { return ctx ? (size_t)4 * 4 * ctx->Hl : 0; }

// one layer's (h, c) of all chains of track lane 0: [target][dir][h|c][Hl]
static int stream_layer_copy(umx_hip_ctx *ctx, int layer, float *host, bool to_host)
{
    if (!ctx || !host || layer < 0 || layer > 2)
        return UMX_ERR_ARG;
    hipError_t e = hipStreamSynchronize(ctx->slot[0].stream);
    if (e == hipSuccess && ctx->ph_next < 0) // outside a phased segment other slots may be busy too
        if (int rc = ctx->sync_all())
            return rc;
    const size_t per = (size_t)4 * ctx->Hl;
    for (int tg = 0; tg < 4 && e == hipSuccess; ++tg)
    {
        float *dev = ctx->state + state_off(tg, layer, 0, 0, ctx->Hl);
        e = to_host ? hipMemcpy(host + tg * per, dev, per * sizeof(float), hipMemcpyDeviceToHost)
                    : hipMemcpy(dev, host + tg * per, per * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}
int umx_hip_stream_get_layer(umx_hip_ctx *ctx, int layer, float *host_dst) { return stream_layer_copy(ctx, layer, host_dst, true); }
int umx_hip_stream_set_layer(umx_hip_ctx *ctx, int layer, const float *host_src)
{
    return stream_layer_copy(ctx, layer, const_cast<float *>(host_src), false);
}

int umx_hip_split_inference(umx_hip_ctx *ctx, const float *audio_host, int length, float *const out_host[4],
                            unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    return ctx ? ctx->track(audio_host, length, -1, out_host, flags, progress, progress_user) : UMX_ERR_ARG;
}
int umx_hip_separate_tracks(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *length, const int *shift_offset,
                            float *const *out_host, unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    return ctx ? ctx->tracks(n_tracks, audio_host, length, shift_offset, out_host, flags, progress, progress_user) : UMX_ERR_ARG;
}
int umx_hip_shift_inference(umx_hip_ctx *ctx, const float *audio_host, int length, int offset, float *const out_host[4],
                            unsigned flags, void (*progress)(float, void *), void *progress_user)
{
    if (!ctx)
        return UMX_ERR_ARG;
    if (offset < 0)
        offset = UMX_REFERENCE_SHIFT; // umx.cpp:115: rand() % 22050, never seeded in the reference (see umx_hip.h)
    return ctx->track(audio_host, length, offset, out_host, flags, progress, progress_user);
}

// debugging: queue `launches` LDS-guard kernels on a private stream (they run beside whatever the caller
// queues next); read the counters back with launches == 0 (returns words corrupted, events in out2[0..1])
int umx_hip_debug_lds_guard(umx_hip_ctx *ctx, int launches, int rounds, unsigned *out2)
{
    static hipStream_t gs = nullptr;
    static unsigned *errs = nullptr;
    if (!ctx)
        return UMX_ERR_ARG;
    if (!gs)
    {
        if (hipStreamCreateWithFlags(&gs, hipStreamNonBlocking) != hipSuccess || hipMalloc(&errs, 8) != hipSuccess)
            return UMX_ERR_HIP;
        (void)hipMemset(errs, 0, 8);
    }
    for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(lds_guard_kernel, dim3(512), dim3(256), 0, gs, errs, rounds);
    if (launches == 0 && out2)
    {
        (void)hipStreamSynchronize(gs);
        (void)hipMemcpy(out2, errs, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(errs, 0, 8);
    }
    return UMX_OK;
}

int umx_hip_segment_begin(umx_hip_ctx *ctx, const float *audio_host, int n, unsigned flags)
{
    return ctx ? ctx->phase_begin(audio_host, n, flags) : UMX_ERR_ARG;
}
int umx_hip_segment_lstm_layer(umx_hip_ctx *ctx, int layer) { return ctx ? ctx->phase_layer(layer) : UMX_ERR_ARG; }
int umx_hip_segment_end(umx_hip_ctx *ctx, float *const out_host[4]) { return ctx ? ctx->phase_end(out_host) : UMX_ERR_ARG; }

void *umx_hip_phase_stream(umx_hip_ctx *ctx) { return ctx ? (void *)ctx->slot[0].stream : nullptr; }
float *umx_hip_stream_state_device(umx_hip_ctx *ctx) { return ctx ? ctx->state : nullptr; }
int umx_hip_segment_begin_device(umx_hip_ctx *ctx, const float *audio_dev, int n, unsigned flags)
{
    return ctx ? ctx->phase_begin_device(audio_dev, n, flags) : UMX_ERR_ARG;
}
int umx_hip_segment_end_device(umx_hip_ctx *ctx, float *const out_dev[4]) { return ctx ? ctx->phase_end_device(out_dev) : UMX_ERR_ARG; }
int umx_hip_segment_masks_device(umx_hip_ctx *ctx) { return ctx ? ctx->phase_masks() : UMX_ERR_ARG; }
int umx_hip_segment_discard(umx_hip_ctx *ctx)
{
    if (!ctx)
        return UMX_ERR_ARG;
    ctx->ph_next = -1; // whatever was queued runs to its end; the next segment may begin
    ctx->cur = 0;
    ctx->clear_used();
    return UMX_OK;
}
int umx_hip_segment_finish_device(umx_hip_ctx *ctx, float *const out_dev[4]) { return ctx ? ctx->phase_finish_device(out_dev) : UMX_ERR_ARG; }
float *umx_hip_target_mag_device(umx_hip_ctx *ctx, int target, size_t *floats)
{
    if (!ctx || target < 0 || target > 3)
        return nullptr;
    if (floats)
        *floats = (size_t)2 * ctx->T * MAGP;
    return ctx->slot[0].lane[0].ta[target].mag;
}
int umx_hip_gate_reserve(int device, int cus)
{
    if (device < 0)
        return UMX_ERR_ARG;
    LstmGate &g = g_gate[device & 15];
    std::lock_guard<std::mutex> lock(g.m);
    if (cus > 0)
        g.reservations.push_back(cus);
    else if (cus < 0) // gives ONE request of that size back: another driver's request on the same device stays
    {
        auto it = std::find(g.reservations.begin(), g.reservations.end(), -cus);
        if (it == g.reservations.end())
            return UMX_ERR_ARG;
        g.reservations.erase(it);
    }
    else
        g.reservations.clear();
    g.reserved = g.reservations.empty() ? 0 : 2 * *std::max_element(g.reservations.begin(), g.reservations.end());
    return UMX_OK;
}

static Stems4 stems4(float *const p[4])
{
    Stems4 s;
    for (int t = 0; t < 4; ++t)
        s.p[t] = reinterpret_cast<float2 *>(p[t]);
    return s;
}
int umx_hip_weight_stems_device(umx_hip_ctx *ctx, float *const stems_dev[4], int n, void *hip_stream)
{
    if (!ctx || !stems_dev || n < 1)
        return UMX_ERR_ARG;
    hipLaunchKernelGGL(track_weight_kernel, dim3((n + 255) / 256, 4), dim3(256), 0, (hipStream_t)hip_stream, stems4(stems_dev), n, ctx->N);
    return hipGetLastError() == hipSuccess ? UMX_OK : UMX_ERR_HIP;
}
int umx_hip_track_accumulate_device(umx_hip_ctx *ctx, float *const track_dev[4], float *sum_weight_dev,
                                    const float *const weighted_dev[4], int offset, int n, void *hip_stream)
{
    if (!ctx || !track_dev || !sum_weight_dev || !weighted_dev || n < 1 || offset < 0)
        return UMX_ERR_ARG;
    float *w[4] = {const_cast<float *>(weighted_dev[0]), const_cast<float *>(weighted_dev[1]), const_cast<float *>(weighted_dev[2]),
                   const_cast<float *>(weighted_dev[3])};
    hipLaunchKernelGGL(track_add_weighted_kernel, dim3((n + 255) / 256, 4), dim3(256), 0, (hipStream_t)hip_stream, stems4(track_dev),
                       sum_weight_dev, stems4(w), offset, n, ctx->N);
    return hipGetLastError() == hipSuccess ? UMX_OK : UMX_ERR_HIP;
}
int umx_hip_track_normalise_device(umx_hip_ctx *ctx, float *const track_dev[4], const float *sum_weight_dev, int length,
                                   void *hip_stream)
{
    if (!ctx || !track_dev || !sum_weight_dev || length < 1)
        return UMX_ERR_ARG;
    hipLaunchKernelGGL(track_normalise_kernel, dim3((length + 255) / 256, 4), dim3(256), 0, (hipStream_t)hip_stream, stems4(track_dev),
                       sum_weight_dev, 0, length);
    return hipGetLastError() == hipSuccess ? UMX_OK : UMX_ERR_HIP;
}

int umx_hip_infer_segment_device(umx_hip_ctx *ctx, const float *audio_dev, int n, float *const out_dev[4],
                                 unsigned flags)
{
    if (!ctx || !out_dev)
        return UMX_ERR_ARG;
    return ctx->infer_device(audio_dev, n, out_dev, flags);
}

int umx_hip_infer_batch_device(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_dev, const int *n,
                               float *const *out_dev, unsigned flags)
{
    if (!ctx)
        return UMX_ERR_ARG;
    return ctx->infer_batch(n_tracks, audio_dev, n, out_dev, flags);
}

// Ordering against the caller's own HIP streams (the engine alternates between two internal streams)
int umx_hip_order_after(umx_hip_ctx *ctx, void *hip_stream)
{
    if (!ctx)
        return UMX_ERR_ARG;
    hipError_t e = hipSuccess;
    if (!ctx->order_ev)
        e = hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventRecord(ctx->order_ev, (hipStream_t)hip_stream);
    for (int si = 0; si < ctx->nslots && e == hipSuccess; ++si)
        e = hipStreamWaitEvent(ctx->slot[si].stream, ctx->order_ev, 0);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}
int umx_hip_order_before(umx_hip_ctx *ctx, void *hip_stream)
{
    if (!ctx)
        return UMX_ERR_ARG;
    // the caller may recycle the buffers of the calls queued so far once its stream is ordered behind them: they can no
    // longer be replayed, so a timeout among them ends as UMX_ERR_TIMEOUT at the next umx_hip_sync (and the log of
    // queued calls stops growing for callers that never call umx_hip_sync)
    if (!ctx->pending.empty())
    {
        ctx->pending_lost = true;
        ctx->pending.clear();
    }
    hipError_t e = hipSuccess;
    if (!ctx->order_ev)
        e = hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming);
    for (int si = 0; si < ctx->nslots && e == hipSuccess; ++si)
    {
        e = hipEventRecord(ctx->order_ev, ctx->slot[si].stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent((hipStream_t)hip_stream, ctx->order_ev, 0);
    }
    // the host-pointer calls download their stems on a stream of their own: "everything queued so far" includes those copies
    if (ctx->copy_stream && e == hipSuccess)
    {
        e = hipEventRecord(ctx->order_ev, ctx->copy_stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent((hipStream_t)hip_stream, ctx->order_ev, 0);
    }
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return UMX_ERR_HIP;
    }
    return UMX_OK;
}

int umx_hip_sync(umx_hip_ctx *ctx)
{
    if (!ctx)
        return UMX_ERR_ARG;
    if (int rc = ctx->sync_all())
        return rc;
    for (int si = 0; si < ctx->nslots; ++si)
    {
        unsigned *dev_status = ctx->slot[si].status;
        if (!dev_status)
            continue;
        unsigned st = 0;
        hipError_t e = hipMemcpy(&st, dev_status, sizeof st, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
        {
            ctx->set_error(hipGetErrorString(e));
            return UMX_ERR_HIP;
        }
        if (st != 0)
        {
            const std::string what = st == 0x80000000u
                                         ? std::string("persistent LSTM kernel: grid barrier timed out (grid not co-resident)")
                                         : "persistent LSTM kernel timed out waiting for a hidden-state granule (code " + std::to_string(st) + ")";
            for (int sj = 0; sj < ctx->nslots; ++sj)
                (void)hipMemset(ctx->slot[sj].status, 0, sizeof(unsigned));
            (void)hipStreamSynchronize(nullptr); // (null-stream memsets are not ordered against the slots' non-blocking streams)
            ctx->persistent_ok = false; // later launches use the per-step driver
            const size_t ncalls = ctx->pending.size();
            if (!ctx->no_recovery && !ctx->pending_lost && ncalls >= 1 && ncalls <= (size_t)umx_hip_ctx::kBackupCalls)
            {
                const int rc = ctx->recover();
                if (rc == UMX_OK)
                {
                    ctx->set_error("recovered: " + what + "; " + std::to_string(ncalls) + " queued segment call(s) were run again with the "
                                   "per-step LSTM driver (bit-identical), which later calls of this context use as well");
                    return UMX_OK;
                }
            }
            // no way back: the aborted launch left a mix of updated and stale chains behind
            (void)hipMemset(ctx->state, 0, sizeof(float) * ctx->state_floats() * ctx->B);
            (void)hipStreamSynchronize(nullptr);
            ctx->clear_used();
            ctx->pending.clear();
            ctx->pending_lost = false;
            ctx->set_error(what + "; the streaming LSTM state was reset to zero, later segments run the per-step driver");
            return UMX_ERR_TIMEOUT;
        }
    }
    ctx->pending.clear();
    ctx->pending_lost = false;
    return UMX_OK;
}

// Host-pointer forms.  H2D, kernels and D2H are queued on the stream of the pipeline slot the segment runs in, so
// with PINNED host buffers consecutive _async calls overlap one segment's transfers with the other's kernels.
int umx_hip_infer_batch_async(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *n,
                              float *const *out_host, unsigned flags)
{
    if (!ctx || !audio_host || !n || !out_host || n_tracks < 1 || n_tracks > ctx->B)
    {
        if (ctx)
            ctx->set_error("infer: bad arguments");
        return UMX_ERR_ARG;
    }
    const int si = ctx->next_slot();
    if (int rc = ctx->ensure_staging())
        return rc;
    Slot &sl = ctx->slot[si];
    hipStream_t st = sl.stream;
    const float *ain[LSTMB_MAX_TRACKS] = {};
    for (int ln = 0; ln < n_tracks; ++ln)
    {
        if (!audio_host[ln])
            continue;
        if (n[ln] < 1 || n[ln] > ctx->N)
        {
            ctx->set_error("infer_segment: need 1 <= n <= segment_samples");
            return UMX_ERR_ARG;
        }
        float *dst = ctx->stage_in[si] + (size_t)2 * ctx->N * ln;
        hipError_t e = hipMemcpyAsync(dst, audio_host[ln], sizeof(float) * 2 * (size_t)n[ln], hipMemcpyHostToDevice, st);
        if (e != hipSuccess)
        {
            ctx->set_error(hipGetErrorString(e));
            return UMX_ERR_HIP;
        }
        ain[ln] = dst;
    }
    if (int rc = ctx->infer_batch(n_tracks, ain, n, ctx->stage_out[si], flags))
        return rc;
    UMX_HIP_CHECK_CTX(ctx, hipEventRecord(sl.k_done, st));
    if (!ctx->pending_lost && !ctx->pending.empty())
    {
        for (int k = 0; k < 4 * n_tracks; ++k)
            ctx->pending.back().host_out[k] = out_host[k];
        for (int ln = 0; ln < n_tracks; ++ln)
            ctx->pending.back().host_audio[ln] = ain[ln] ? audio_host[ln] : nullptr;
    }
    // the stems go out on the copy stream as soon as they are complete, beside whatever runs next (see queue_download)
    umx_hip_ctx::DeferredDownload d;
    d.valid = true;
    d.si = si;
    d.nb = n_tracks;
    for (int ln = 0; ln < n_tracks; ++ln)
    {
        d.n[ln] = ain[ln] ? n[ln] : 0;
        for (int s = 0; s < 4; ++s)
            d.host[4 * ln + s] = out_host[4 * ln + s];
    }
    return ctx->queue_download(d, ctx->copy_stream);
}

int umx_hip_infer_batch(umx_hip_ctx *ctx, int n_tracks, const float *const *audio_host, const int *n, float *const *out_host,
                        unsigned flags)
{
    if (int rc = umx_hip_infer_batch_async(ctx, n_tracks, audio_host, n, out_host, flags))
        return rc;
    return umx_hip_sync(ctx);
}

int umx_hip_infer_segment_async(umx_hip_ctx *ctx, const float *audio_host, int n, float *const out_host[4], unsigned flags)
{
    if (!ctx || !audio_host || !out_host)
    {
        if (ctx)
            ctx->set_error("infer_segment: bad arguments");
        return UMX_ERR_ARG;
    }
    return umx_hip_infer_batch_async(ctx, 1, &audio_host, &n, out_host, flags);
}

int umx_hip_infer_segment(umx_hip_ctx *ctx, const float *audio_host, int n, float *const out_host[4], unsigned flags)
{
    if (int rc = umx_hip_infer_segment_async(ctx, audio_host, n, out_host, flags))
        return rc;
    return umx_hip_sync(ctx); // umx_inference returns its outputs: synchronous
}

void *umx_hip_stream_handle(umx_hip_ctx *ctx) { return ctx ? (void *)ctx->slot[ctx->cur].stream : nullptr; }
int umx_hip_nb_frames(const umx_hip_ctx *ctx) { return ctx ? ctx->T : 0; }
int umx_hip_segment_samples(const umx_hip_ctx *ctx) { return ctx ? ctx->N : 0; }
int umx_hip_hidden(const umx_hip_ctx *ctx) { return ctx ? ctx->H : 0; }

long umx_hip_read_tap(umx_hip_ctx *ctx, const char *what, int target, float *dst, size_t cap)
{
    if (!ctx || !what || target < 0 || target > 3)
        return -1;
    std::string w = what;
    const int T = ctx->T, H = ctx->H;
    int which = ctx->cur, lane = 0;
    {
        const size_t hash = w.find('#'); // "name#k": track lane k (default 0)
        if (hash != std::string::npos)
        {
            lane = atoi(w.c_str() + hash + 1);
            const size_t at = w.find('@', hash);
            w = w.substr(0, hash) + (at == std::string::npos ? "" : w.substr(at));
            if (lane < 0 || lane >= ctx->B)
                return -1;
        }
    }
    if (w.size() > 2 && w[w.size() - 2] == '@') // "name@s": pipeline slot s instead of the most recent one
    {
        which = w.back() - '0';
        w = w.substr(0, w.size() - 2);
        if (which < 0 || which >= ctx->nslots)
            return -1;
    }
    const Lane &sl = ctx->slot[which].lane[lane];
    const void *src = nullptr;
    size_t nfl = 0, src_ld = 0, rows = 0, cols = 0; // strided copy when src_ld != cols
    int computed = 0; // 1 |X|, 2 mask x |X|, 3 mask as (T, 4098): formed by a tap kernel into ctx->tap_tmp
    if (w == "spec") { src = sl.spec; nfl = (size_t)2 * 2 * T * NBINS; }
    else if (w == "mix_mag") { src = sl.spec; computed = 1; nfl = (size_t)2 * T * NBINS; }
    else if (w == "x") { src = sl.x; nfl = (size_t)T * KX; }
    else if (w == "fc1") { src = sl.ta[target].cat; rows = T; cols = H; src_ld = 2 * H; nfl = rows * cols; }
    else if (w == "lstm") { src = sl.ta[target].cat + H; rows = T; cols = H; src_ld = 2 * H; nfl = rows * cols; }
    else if (w == "lstm_l0") { src = sl.ta[target].la; nfl = (size_t)T * H; }
    else if (w == "lstm_l1") { src = sl.ta[target].lb; nfl = (size_t)T * H; }
    else if (w == "proj") { src = sl.ta[target].P; nfl = (size_t)T * 4 * H; }
    else if (w == "fc2") { src = sl.ta[target].a2; nfl = (size_t)T * H; }
    else if (w == "mask") { src = sl.ta[target].mag; computed = 3; nfl = (size_t)T * NOUT; }
    else if (w == "target_mag") { src = sl.ta[target].mag; computed = 2; nfl = (size_t)2 * T * NBINS; }
    else if (w == "y") { src = sl.y + (size_t)target * 2 * T * NBINS; nfl = (size_t)2 * 2 * T * NBINS; }
    else if (w == "max_abs") { src = sl.maxabs; nfl = 1; }
    else return -1;
    if (!src)
        return -2;
    if (!dst)
        return (long)nfl;
    if (cap < nfl)
        return -3;
    if (ctx->sync_all() != UMX_OK)
        return -4;
    if (computed)
    {
        if (!ctx->tap_tmp && ctx->dalloc(&ctx->tap_tmp, (size_t)2 * T * NBINS) != UMX_OK)
            return -4;
        const unsigned blocks = (unsigned)((nfl + 255) / 256);
        if (computed == 1)
            hipLaunchKernelGGL(tap_mix_mag_kernel, dim3(blocks), dim3(256), 0, nullptr, sl.spec, nfl, ctx->tap_tmp);
        else if (computed == 2)
            hipLaunchKernelGGL(tap_target_mag_kernel, dim3(blocks), dim3(256), 0, nullptr, sl.spec, sl.ta[target].mag, nfl, ctx->tap_tmp);
        else
            hipLaunchKernelGGL(tap_mask_kernel, dim3(blocks), dim3(256), 0, nullptr, sl.ta[target].mag, T, ctx->tap_tmp);
        if (hipDeviceSynchronize() != hipSuccess)
            return -4;
        src = ctx->tap_tmp;
    }
    hipError_t e;
    if (rows)
        e = hipMemcpy2D(dst, cols * sizeof(float), src, src_ld * sizeof(float), cols * sizeof(float), rows,
                        hipMemcpyDeviceToHost);
    else
        e = hipMemcpy(dst, src, nfl * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
    {
        ctx->set_error(hipGetErrorString(e));
        return -4;
    }
    if (w == "max_abs")
    {
        unsigned bits;
        memcpy(&bits, dst, 4);
        float m;
        memcpy(&m, &bits, 4);
        dst[0] = std::max(1.0f, m / WIENER_SCALE);
    }
    return (long)nfl;
}

int umx_hip_stage_times_slot(umx_hip_ctx *ctx, int slot_index, const char **names, float *ms, int cap);
int umx_hip_stage_times(umx_hip_ctx *ctx, const char **names, float *ms, int cap)
{
    return ctx ? umx_hip_stage_times_slot(ctx, ctx->cur, names, ms, cap) : 0;
}

int umx_hip_stage_times_slot(umx_hip_ctx *ctx, int slot_index, const char **names, float *ms, int cap)
{
    if (!ctx || slot_index < 0 || slot_index >= ctx->nslots)
        return 0;
    Slot &sl = ctx->slot[slot_index];
    if (!sl.have_times || ctx->sync_all() != UMX_OK)
        return 0;
    int n = std::min(cap, (int)ST_COUNT);
    for (int i = 0; i < n; ++i)
    {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, sl.ev[i], sl.ev[i + 1]);
        if (names)
            names[i] = kStageNames[i];
        if (ms)
            ms[i] = t;
    }
    return ST_COUNT;
}

// per stage: the time from the event recorded BEHIND the stage's split kernel to the next stage's event, i.e. the stage's main
// kernel alone (GEMM stages of plane contexts); the stage time where there is no such event.  Same conventions as umx_hip_stage_times_slot.
int umx_hip_stage_kernel_times_slot(umx_hip_ctx *ctx, int slot_index, float *ms, int cap)
{
    if (ctx && slot_index < 0)
        slot_index = ctx->cur;
    if (!ctx || slot_index >= ctx->nslots)
        return 0;
    Slot &sl = ctx->slot[slot_index];
    if (!sl.have_times || ctx->sync_all() != UMX_OK)
        return 0;
    const int n = std::min(cap, (int)ST_COUNT);
    for (int i = 0; i < n; ++i)
    {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, sl.evk_set[i] ? sl.evk[i] : sl.ev[i], sl.ev[i + 1]);
        if (ms)
            ms[i] = t;
    }
    return ST_COUNT;
}

int umx_hip_lstm_was_persistent(const umx_hip_ctx *ctx) { return ctx && ctx->slot[ctx->cur].last_persistent ? 1 : 0; }

int umx_hip_lstm_mode(umx_hip_ctx *ctx)
{
    if (!ctx || !ctx->slot[ctx->cur].last_persistent)
        return 0;
    unsigned st[2] = {0, 0};
    const unsigned *src = ctx->slot[ctx->cur].status;
    if (ctx->sync_all() != UMX_OK || hipMemcpy(st, src, sizeof st, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return st[1] ? 2 : 1;
}

int umx_hip_debug_lstm_profile(umx_hip_ctx *ctx, unsigned long long *out48)
{
    if (!ctx || !out48)
        return UMX_ERR_ARG;
    if (ctx->sync_all() != UMX_OK ||
        hipMemcpy(out48, ctx->slot[ctx->cur].lprof, sizeof(unsigned long long) * 48, hipMemcpyDeviceToHost) != hipSuccess)
        return UMX_ERR_HIP;
    return UMX_OK;
}

// where the workgroups of the last profiled one-track recurrence launch ran: out[i] = xcc << 48 | chain << 40 | slice << 32 | HW_ID
int umx_hip_debug_lstm_placement(umx_hip_ctx *ctx, unsigned long long *out, int n)
{
    if (!ctx || !out || n < 0 || n > 960 + 64 * 8 * 5)
        return UMX_ERR_ARG;
    if (ctx->sync_all() != UMX_OK ||
        hipMemcpy(out, ctx->slot[ctx->cur].lprof + 64, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost) != hipSuccess)
        return UMX_ERR_HIP;
    return UMX_OK;
}

} // extern "C"
